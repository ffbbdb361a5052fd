cd shot_vae_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
mkdir -p ../../build/ab
/opt/rocm/bin/hipcc $FLAGS -DSV_X3_MODES=2 -c conv3x3x.hip -o ../../build/ab/x3m.o 2>/dev/null
OBJS=""; for o in igemm halo hwgrad conv3x3 conv3x3w wgrad wgrad3x3 small runtime; do OBJS="$OBJS $o.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../build/ab/x3m.o $OBJS -o ../../build/ab/lib_x3m.so
cd ../..
export SV_LIB_PATH=$PWD/build/ab/lib_x3m.so
python tools/probes/x3m_where.py 2>&1 | tail -12

python tools/layer_bench.py 2>&1 | grep "of bf16" | head -9
