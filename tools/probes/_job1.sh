SV_BWDG_TAGS="base la64 h80 h64 w80" bash tools/probes/bwdg_ablate.sh run
cd /tmp && export TMPDIR=/tmp
for t in base la64 h80 h64 w80; do
  echo "== $t"; SV_LIB_PATH=/root/repo/build/ab/bwdg_$t.so SV_BENCH_FUSED_BLOCKS=248 python3 /root/repo/tools/pmc_sq.py 2048 64 16 64 bwd2 2>&1 | grep "LDS_BANK_CONFLICT\|SQ_WAIT_INST_LDS\|SQ_ACTIVE_INST_LDS"
done
