#!/bin/bash
# scratch driver of a GPU-box visit (round 6); edited per visit
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; OUT=$R/gpurun_out/r6av; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for v in on off; do
  if [ $v = off ]; then export RPG_HIP_LIB=$R/relpose-gnn_amd/lib/librelpose_gnn_hip_probe.so; fi
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_$v -o p -- python3 $R/tools/conv_bench.py --bf16 --block64 --nimg 512 --reps 3 > $OUT/pmc_$v.log 2>&1
  python3 - $OUT/pmc_$v $v <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "block64" in r["Kernel_Name"]:
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("parity", sys.argv[2], k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "dispatches", len(next(iter(d.values()))))
PY
  rm -rf $OUT/pmc_$v
done 2>&1 | tee $OUT/block_parity_pmc.txt
