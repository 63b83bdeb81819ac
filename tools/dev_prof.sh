set -x
cd /root/repo
python bench.py --steps 30 --warmup 5 --ppl 1,1 > gpurun_out/bench_11.json 2> gpurun_out/bench.err
cat gpurun_out/bench_11.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_r01 -o r01 -- python /root/repo/bench.py --steps 20 --warmup 5 --ppl 1,1 --no-cpu-baseline > /root/repo/gpurun_out/prof_bench.json 2> /root/repo/gpurun_out/prof.err
ls -R /root/repo/gpurun_out/prof_r01 | head -30
