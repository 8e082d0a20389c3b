#!/bin/bash
# Round-4 GPU-box visits: tools/r4_visit.sh <tag> <steps...>
TAG=${1:-v}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
for step in "$@"; do
  case $step in
    tests)
      rm -f gpurun_out/parity_report.jsonl
      timeout 3000 python -m pytest tests -q -m gpu > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest.log"
      tail -25 "$OUT/pytest.log"; cp gpurun_out/parity_report.jsonl "$OUT/" 2>/dev/null;;
    newtests)
      rm -f gpurun_out/parity_report.jsonl
      timeout 1800 python -m pytest tests/test_hip_eval_geometry.py "tests/test_hip_model.py::test_reference_written_files_to_forward_g9" "tests/test_hip_model.py::test_eval_stream_input_pipeline_variants_agree" "tests/test_hip_model.py::test_bf16_encoder_takes_host_rounded_bf16_images" "tests/test_hip_model.py::test_gnn_fused_aggregation_equals_reference_order" -q -m gpu --durations=8 > "$OUT/pytest_new.log" 2>&1; echo "pytest rc=$?" | tee -a "$OUT/pytest_new.log"
      tail -40 "$OUT/pytest_new.log"; cp gpurun_out/parity_report.jsonl "$OUT/parity_new.jsonl" 2>/dev/null; cat "$OUT/parity_new.jsonl";;
    bench)
      timeout 1200 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"; cut -c1-6000 "$OUT/bench.json";;
    benchq)
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs > "$OUT/benchq.json" 2> "$OUT/benchq.err"; echo "benchq rc=$?"; cut -c1-700 "$OUT/benchq.json";;
    lat1)
      timeout 300 python bench.py --steps 40 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --graphs 1 --streams 1 --no-kernel-timing 2>/dev/null | cut -c1-300;;
    prof1)
      export TMPDIR=/tmp; cd /tmp
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_b1" -o t -- python3 "$R/bench.py" --steps 30 --warmup 5 --cpu-baseline-seconds 0 --no-other-configs --graphs 1 --streams 1 --no-kernel-timing > "$OUT/trace_b1.log" 2>&1
      cd "$R"; f=$(ls "$OUT"/trace_b1/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 tools/rocprof_summary.py "$f" "$OUT/kernel_stats_b1.txt" > /dev/null; head -40 "$OUT/kernel_stats_b1.txt"; rm -f "$OUT"/trace_b1/*kernel_trace.csv
      tail -2 "$OUT/trace_b1.log" | cut -c1-300;;
    convbench)
      timeout 600 python tools/conv_bench.py --bf16 --nimg 512 --warm 3 --reps 10 --only l > "$OUT/conv_bf16_512.txt" 2>&1; cat "$OUT/conv_bf16_512.txt"
      timeout 600 python tools/conv_bench.py --bf16 --nimg 256 --warm 3 --reps 10 --only l > "$OUT/conv_bf16_256.txt" 2>&1; cat "$OUT/conv_bf16_256.txt";;
    convbench32)
      timeout 600 python tools/conv_bench.py --nimg 256 --warm 3 --reps 10 > "$OUT/conv_f32_256.txt" 2>&1; cat "$OUT/conv_f32_256.txt";;
    benchbf16)
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --graphs 64 --encoder-dtype bf16 > "$OUT/bench_bf16.json" 2> "$OUT/bench_bf16.err"; echo "rc=$?"; cut -c1-1200 "$OUT/bench_bf16.json"
      timeout 600 python bench.py --steps 20 --warmup 5 --cpu-baseline-seconds 0 --graphs 64 --encoder-dtype bf16 --gnn-dtype bf16 > "$OUT/bench_bf16_gnn.json" 2> "$OUT/bench_bf16_gnn.err"; echo "rc=$?"; cut -c1-1200 "$OUT/bench_bf16_gnn.json";;
    custom)
      bash "$R/tools/r4_custom.sh" "$OUT";;
    *) echo "unknown step $step";;
  esac
done
