cd shot_vae_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
mkdir -p ../../build/ab
OBJS=""; for o in igemm halo hwgrad conv3x3 conv3x3w wgrad wgrad3x3 small runtime; do OBJS="$OBJS $o.o"; done
for v in 1 2 3; do
  /opt/rocm/bin/hipcc $FLAGS -DSV_X3_EPD=$v -c conv3x3x.hip -o ../../build/ab/x3e$v.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../build/ab/x3e$v.o $OBJS -o ../../build/ab/lib_x3e$v.so &
done
wait
cd ../..
for v in 1 2 3; do
  echo "== EPD=$v"
  SV_LIB_PATH=$PWD/build/ab/lib_x3e$v.so python tools/layer_bench.py 2>&1 | grep "of bf16" | grep -v wgrad
done
