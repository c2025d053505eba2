# usage: bash tools/profile_full.sh <tag> [pytest|nopytest]  -- GPU box: whole GPU suite, default bench, rocprof stats + serial
# timeline of the same command, PMC passes (separate), then the 1S+1T pipeline with both ranks on the one GPU
tag=$1; what=${2:-pytest}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "$what" = pytest ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest.log 2>&1; rc=$?
  echo "pytest rc=$rc"; tail -5 gpurun_out/${tag}_pytest.log
  [ $rc -ne 0 ] && exit $rc
fi
timeout -k 10 500 python3 bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err; rc=$?
echo "bench rc=$rc"; [ $rc -ne 0 ] && { tail -c 2000 gpurun_out/${tag}_bench_default.err; exit $rc; }
python3 tools/show_bench.py gpurun_out/${tag}_bench_default.json | grep -E '"value"|ms_per_step|frac|edges_per_s|hit_rate|GBps' | head -40
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc1 -- python3 bench.py --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_default.log 2>&1 || exit 1
python3 tools/stats_summary.py gpurun_out/pc1 > gpurun_out/${tag}_default_stats.md
cp $(find gpurun_out/pc1 -name "*kernel_stats.csv") gpurun_out/${tag}_bench_default_kernel_stats.csv
python3 tools/overlap_stats.py gpurun_out/pc1 > gpurun_out/${tag}_overlap.txt 2>&1
rm -rf gpurun_out/pc1
head -25 gpurun_out/${tag}_default_stats.md
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pc2 -- python3 bench.py --steps 40 --warmup 5 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/${tag}_prof_serial.log 2>&1 || exit 1
python3 tools/chain_timeline.py gpurun_out/pc2 20 > gpurun_out/${tag}_timeline_serial.txt 2>&1; rm -rf gpurun_out/pc2
cat gpurun_out/${tag}_timeline_serial.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_$c -- python3 bench.py --steps 20 --warmup 3 --no-overlap --no-cpu-baseline --timed-only > gpurun_out/${tag}_pmc_$c.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/${tag}_pmc_FETCH_SIZE.log > gpurun_out/${tag}_pmc_traffic.json 2> gpurun_out/${tag}_pmc.err
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
head -12 gpurun_out/${tag}_pmc_traffic.json; tail -3 gpurun_out/${tag}_pmc.err
# config 3's trainer-side leg from KERNEL durations: (a) bench.py's roofline_extract (this GPU samples too), (b) the
# trainer process of the decoupled 1S+1T run (the trainer drains a full queue alone); rocprofv3's preloaded library is
# inherited by the rank processes bench.py starts, one trace per process
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/px1 -- python3 bench.py --steps 64 --warmup 5 --windows 1 --no-train-leg --no-cpu-baseline --presample-variants "" > gpurun_out/${tag}_prof_extract_leg.log 2>&1 || exit 1
python3 - <<P > gpurun_out/${tag}_extract_kernel_stats.md
import json, subprocess
l = json.loads([x for x in open("gpurun_out/${tag}_prof_extract_leg.log") if x.startswith("{")][-1])
r = l["roofline_extract"]
print("## roofline_extract under rocprofv3 --kernel-trace (presample_epoch %d, hit rate %.4f, %.4f ms per batch in this profiled run)" % (r["presample_epoch"], r["hit_rate"], r["ms_per_step"]))
print(subprocess.run(["python3", "tools/extract_kernel_rates.py", "gpurun_out/px1", str(r["miss"]["bytes_per_step"]), str(r["cached"]["bytes_per_step"])], capture_output=True, text=True).stdout)
P
python3 tools/stats_summary.py gpurun_out/px1 >> gpurun_out/${tag}_extract_kernel_stats.md; rm -rf gpurun_out/px1
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/px2 -- python3 bench.py --gpus 2 --decoupled --no-train-leg --no-cpu-baseline > gpurun_out/${tag}_prof_decoupled.log 2>&1 || exit 1
python3 - <<P >> gpurun_out/${tag}_extract_kernel_stats.md
import json, subprocess
l = json.loads([x for x in open("gpurun_out/${tag}_prof_decoupled.log") if x.startswith("{")][-1])
pl = l["pipeline"]
rows = l["input_nodes_per_step"]; miss_b = pl["miss"]["bytes_per_step"]; hit_b = (rows - miss_b / 512.0) * (2 * 512 + 8)
print("\n## the trainer PROCESS of bench.py --gpus 2 --decoupled under rocprofv3 --kernel-trace (it drains a full queue alone; %.4f ms per batch, second half %.4f)" % (l["ms_per_step"], pl["consumed_second_half_ms_per_batch"]))
print(subprocess.run(["python3", "tools/extract_kernel_rates.py", "gpurun_out/px2", str(miss_b), str(hit_b)], capture_output=True, text=True).stdout)
P
rm -rf gpurun_out/px2; cat gpurun_out/${tag}_extract_kernel_stats.md
timeout -k 10 500 python3 bench.py --gpus 2 --no-cpu-baseline > gpurun_out/${tag}_bench_gpus2.json 2> gpurun_out/${tag}_bench_gpus2.err; echo "bench2 rc=$?"
tail -c 800 gpurun_out/${tag}_bench_gpus2.err
python3 tools/show_bench.py gpurun_out/${tag}_bench_gpus2.json | head -60
