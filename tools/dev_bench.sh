set -x
cd /root/repo
python tools/dev_knn.py > gpurun_out/knn.log 2>&1
python bench.py --steps 30 --warmup 5 > gpurun_out/bench_44.json 2> gpurun_out/bench_44.err
for ppl in 1,1 2,2 4,2 2,4 1,4 4,1 1,2 2,1; do
  python bench.py --steps 20 --warmup 5 --ppl $ppl --no-cpu-baseline > gpurun_out/bench_$ppl.json 2>> gpurun_out/bench_44.err
done
cat gpurun_out/knn.log
tail -5 gpurun_out/bench_44.err
for f in gpurun_out/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1], "value %.3e ms/step %.3f" % (d["value"], d["ms_per_step"]), d["config"]["ppl"], d["kernel_us"], "roof", d["roofline"]["achieved"], "cpu", d.get("cpu_baseline",{}).get("value"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
