# A/B of two builds of the library on one box, interleaved: tools/ubench/ab.sh <pool 0|1>
for rep in 1 2; do
for v in A B; do
  echo "== lib$v pool=$1"; AGP_HIP_LIB=$GRAFT_REPO_ROOT/agplace_amd/lib/variants/lib$v.so timeout 200 python tools/bblock_bench.py --reps 40 --rounds 2 --pool $1 2>&1 | grep -E "round 1"
done; done
