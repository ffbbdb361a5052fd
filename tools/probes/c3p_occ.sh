cd shot_vae_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
mkdir -p ../../build/ab
OBJS=""; for o in igemm halo hwgrad conv3x3w conv3x3x wgrad wgrad3x3 small runtime; do OBJS="$OBJS $o.o"; done
/opt/rocm/bin/hipcc $FLAGS -DSV_C3P_MODES=1 -c conv3x3.hip -o ../../build/ab/c3m.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../build/ab/c3m.o $OBJS -o ../../build/ab/lib_c3m.so
cd ../..
export SV_LIB_PATH=$PWD/build/ab/lib_c3m.so
for pb in 512 768 1024; do
  echo "== persistent blocks $pb"
  SV_BENCH_PERSISTENT_BLOCKS=$pb python tools/layer_bench.py 2048 32 32 32 fwd dgrad 2>/dev/null | grep "of bf16"
  SV_BENCH_PERSISTENT_BLOCKS=$pb python tools/layer_bench.py 2048 64 16 64 fwd dgrad 2>/dev/null | grep "of bf16"
done
