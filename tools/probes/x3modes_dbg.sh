cd shot_vae_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -I../../include -Wno-unused-function"
mkdir -p ../../build/ab
OBJS=""; for o in igemm halo hwgrad conv3x3 conv3x3w wgrad wgrad3x3 small runtime; do OBJS="$OBJS $o.o"; done
i=0
for v in "-DSV_X3_MODES=2 -DSV_X3_NOP=1" "-DSV_X3_MODES=2 -DSV_X3_NOP=3" "-DSV_X3_MODES=0 -DSV_X3_NOP=1"; do
  i=$((i+1))
  ( /opt/rocm/bin/hipcc $FLAGS $v -c conv3x3x.hip -o ../../build/ab/x3n$i.o 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ../../build/ab/x3n$i.o $OBJS -o ../../build/ab/lib_x3n$i.so ) &
done
wait
cd ../..
i=0
for v in "modes2 nop1" "modes2 nop3" "modes0 nop1"; do
  i=$((i+1))
  echo "== $v"
  export SV_LIB_PATH=$PWD/build/ab/lib_x3n$i.so
  python tools/probes/x3m_where.py 2>&1 | grep "differing\|non-finite\|sums equal"
  python tools/layer_bench.py 2>&1 | grep "of bf16" | grep -v wgrad
done
