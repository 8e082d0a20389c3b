#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on the GPU box (two counter-only passes); prints counter x 1024 per dispatch.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fetch_calib
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/fc "$R/tools/probes/fetch_calib_probe.hip" || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -o p -- /tmp/fc > "$OUT/$c.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys
from collections import defaultdict
root = sys.argv[1]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{root}/{c}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(c, "no csv"); continue
    acc = defaultdict(float)
    order = []
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c:
            continue
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"][:28])
        if key not in acc:
            order.append(key)
        acc[key] += float(r["Counter_Value"])
    for k in order:
        print(f"{c:10s} dispatch {k[0]:3d} {k[1]:30s} counter*1024 = {acc[k]*1024/2**20:10.1f} MiB")
PY
tail -2 "$OUT/FETCH_SIZE.log"
