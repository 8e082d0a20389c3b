#!/bin/bash
# Round-3 GPU-box visits: tools/r3_visit.sh <tag> <steps...>   (steps: probe tests bench evalstream convbench)
TAG=${1:-v}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
for step in "$@"; do
  case $step in
    probe)
      timeout 60 tools/probes/glds_probe.bin > "$OUT/glds_probe.txt" 2>&1; echo "probe rc=$?"; cat "$OUT/glds_probe.txt";;
    tests)
      rm -f gpurun_out/parity_report.jsonl
      timeout 2400 python -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
      tail -25 "$OUT/pytest.log"; cp gpurun_out/parity_report.jsonl "$OUT/" 2>/dev/null;;
    bench)
      timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-3000 "$OUT/bench.json";;
    benchq)
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 > "$OUT/benchq.json" 2> "$OUT/benchq.err"; echo "benchq rc=$?"; cut -c1-600 "$OUT/benchq.json";;
    benchbf16)
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --graphs 64 --encoder-dtype bf16 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; echo "rc=$?"; cut -c1-1500 "$OUT/bench_bf16.json"
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 > "$OUT/bench_bf16_gnn.json" 2> "$OUT/bench_bf16_gnn.err"; echo "rc=$?"; cut -c1-1500 "$OUT/bench_bf16_gnn.json";;
    evalstream)
      for inp in resident host pinned; do
        timeout 600 python tools/eval_stream.py --graphs 2000 --shape 256x341 --input $inp >> "$OUT/eval_stream.jsonl" 2>> "$OUT/eval_stream.err"; echo "eval $inp rc=$?"
      done
      timeout 600 python tools/eval_stream.py --graphs 4000 --shape 256x341 --input host --encoder-dtype bf16 --gnn-dtype bf16 >> "$OUT/eval_stream.jsonl" 2>> "$OUT/eval_stream.err"
      timeout 600 python tools/eval_stream.py --graphs 4000 --shape 256x341 --input resident --encoder-dtype bf16 --gnn-dtype bf16 >> "$OUT/eval_stream.jsonl" 2>> "$OUT/eval_stream.err"
      cat "$OUT/eval_stream.jsonl";;
    convbench)
      timeout 600 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --reps 10 --only l > "$OUT/conv_bf16_512.txt" 2>&1; cat "$OUT/conv_bf16_512.txt"
      timeout 600 python tools/conv_bench.py --bf16 --nimg 256 --warm 3 --reps 10 --only l > "$OUT/conv_bf16_256.txt" 2>&1; cat "$OUT/conv_bf16_256.txt";;
    newtests)
      rm -f gpurun_out/parity_report.jsonl
      timeout 1500 python -m pytest tests/test_hip_bf16.py tests/test_hip_bench_geometry.py::test_configs2_bf16_forward_as_benched_vs_oracle tests/test_hip_model.py::test_eval_stream_input_pipeline_variants_agree -q -m gpu > "$OUT/pytest_new.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_new.log"
      tail -40 "$OUT/pytest_new.log"; cp gpurun_out/parity_report.jsonl "$OUT/parity_new.jsonl" 2>/dev/null;;
    dmasweep)
      timeout 900 python tools/conv_bench.py --bf16 --nimg 512 --warm 2 --reps 7 --only l --dma-sweep 0,1,2,3,4,5,6,7,8,9 > "$OUT/dma_sweep_512.txt" 2>&1; cat "$OUT/dma_sweep_512.txt"
      timeout 900 python tools/conv_bench.py --bf16 --nimg 256 --warm 2 --reps 7 --only l --dma-sweep 0,1,3,4,5,7,9 > "$OUT/dma_sweep_256.txt" 2>&1; cat "$OUT/dma_sweep_256.txt";;
    l1sweep)
      timeout 900 python tools/conv_bench.py --bf16 --nimg 512 --warm 2 --reps 7 --only "l1.c" --dma-sweep 5,9,10,11,12,13 > "$OUT/l1_sweep.txt" 2>&1
      timeout 900 python tools/conv_bench.py --bf16 --nimg 256 --warm 2 --reps 7 --only "l1.c" --dma-sweep 5,9,10,11,12,13 >> "$OUT/l1_sweep.txt" 2>&1
      cat "$OUT/l1_sweep.txt";;
    widesweep)
      timeout 900 python tools/conv_bench.py --bf16 --nimg 512 --warm 2 --reps 7 --only "l1.c" --dma-sweep 5,11,12 > "$OUT/wide_sweep.txt" 2>&1
      timeout 900 python tools/conv_bench.py --bf16 --nimg 512 --warm 2 --reps 7 --only "l2." --dma-sweep 7,10 >> "$OUT/wide_sweep.txt" 2>&1
      timeout 900 python tools/conv_bench.py --bf16 --nimg 256 --warm 2 --reps 7 --only "l1.c" --dma-sweep 5,11,12 >> "$OUT/wide_sweep.txt" 2>&1
      timeout 900 python tools/conv_bench.py --bf16 --nimg 256 --warm 2 --reps 7 --only "l2." --dma-sweep 7,10 >> "$OUT/wide_sweep.txt" 2>&1
      cat "$OUT/wide_sweep.txt";;
    lat1)
      for v in 1 6 8 12 16; do
        echo "wino split min steps $v"; timeout 300 python bench.py --steps 40 --warmup 5 --cpu-baseline-seconds 0 --graphs 1 --streams 1 --no-kernel-timing --tune 8=$v 2>/dev/null | cut -c1-200
      done
      for st in 1 2 3 4; do
        echo "bf16 streams $st"; timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 --streams $st --no-kernel-timing 2>/dev/null | cut -c1-200
      done;;
    patchsweep)
      timeout 900 python tools/conv_bench.py --bf16 --nimg 512 --warm 2 --reps 7 --only l --dma-sweep 0,5,7 > "$OUT/patch_sweep_512.txt" 2>&1; cat "$OUT/patch_sweep_512.txt"
      timeout 900 python tools/conv_bench.py --bf16 --nimg 256 --warm 2 --reps 7 --only l --dma-sweep 0,5,7 > "$OUT/patch_sweep_256.txt" 2>&1; cat "$OUT/patch_sweep_256.txt";;
    prof1)
      export TMPDIR=/tmp; cd /tmp
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_b1" -o t -- python3 "$R/bench.py" --steps 30 --warmup 5 --cpu-baseline-seconds 0 --graphs 1 --streams 1 --no-kernel-timing > "$OUT/trace_b1.log" 2>&1
      cd "$R"; f=$(ls "$OUT"/trace_b1/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/rocprof_summary.py "$f" "$OUT/kernel_stats_b1.txt" > /dev/null; head -40 "$OUT/kernel_stats_b1.txt"; rm -f "$OUT"/trace_b1/*kernel_trace.csv
      tail -2 "$OUT/trace_b1.log" | cut -c1-300;;
    evalhost)
      for inp in host resident; do
        timeout 600 python tools/eval_stream.py --graphs 2000 --shape 256x341 --input $inp >> "$OUT/eval_stream.jsonl" 2>> "$OUT/eval_stream.err"; echo "eval $inp rc=$?"
      done
      timeout 600 python tools/eval_stream.py --graphs 4000 --shape 256x341 --input host --encoder-dtype bf16 --gnn-dtype bf16 >> "$OUT/eval_stream.jsonl" 2>> "$OUT/eval_stream.err"
      cat "$OUT/eval_stream.jsonl";;
    stemtest)
      timeout 900 python -m pytest tests/test_hip_bf16.py -q -m gpu -k "stem" > "$OUT/pytest_stem.log" 2>&1; echo "pytest rc=$?"; tail -15 "$OUT/pytest_stem.log"
      timeout 600 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --reps 10 --only stem > "$OUT/stem_512.txt" 2>&1; cat "$OUT/stem_512.txt";;
    stemablate)
      for d in ${STEM_DBG:-0 8 4 12 2 6 14 46 1 16 17 63}; do
        echo "dbg $d"; RPG_STEM_DBG=$d timeout 600 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --reps 10 --only stem 2>&1 | grep fused
      done | tee "$OUT/stem_ablate.txt";;
    ws64)
      timeout 900 python -m pytest tests/test_hip_bf16.py -q -m gpu -x -k "weights_stationary" > "$OUT/pytest_ws64.log" 2>&1; echo "pytest rc=$?"; tail -15 "$OUT/pytest_ws64.log"
      for m in 0 1 2; do echo "ws64 mode $m"; RPG_WS64=$m timeout 600 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --reps 10 --only l1.c 2>&1 | grep conv; done | tee "$OUT/ws64_512.txt";;
    evalbf16)
      timeout 900 python -m pytest tests/test_hip_model.py tests/test_hip_bf16.py -q -m gpu -x -k "host_rounded or fused_stem or eval_stream" > "$OUT/pytest_bfin.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_bfin.log"
      for h2d in bf16 f32; do
        timeout 600 python tools/eval_stream.py --graphs 4000 --shape 256x341 --input host --encoder-dtype bf16 --gnn-dtype bf16 --h2d $h2d >> "$OUT/eval_stream_bf16.jsonl" 2>> "$OUT/eval_stream_bf16.err"
      done
      timeout 600 python tools/eval_stream.py --graphs 4000 --shape 256x341 --input pinned --encoder-dtype bf16 --gnn-dtype bf16 >> "$OUT/eval_stream_bf16.jsonl" 2>> "$OUT/eval_stream_bf16.err"
      timeout 600 python tools/eval_stream.py --graphs 4000 --shape 256x341 --input resident --encoder-dtype bf16 --gnn-dtype bf16 >> "$OUT/eval_stream_bf16.jsonl" 2>> "$OUT/eval_stream_bf16.err"
      cut -c1-60,330-700 "$OUT/eval_stream_bf16.jsonl"; tail -3 "$OUT/eval_stream_bf16.err";;
    *) echo "unknown step $step";;
  esac
done
