set -u
export TMPDIR=/tmp
R=$(pwd)
mkdir -p gpurun_out/r02a
python bench.py --steps 50 --warmup 5 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r02a/rocprof_bench.json 2> $R/gpurun_out/r02a/rocprof.err
F=$(find /tmp/kt -name '*kernel_trace.csv' | head -1)
cp $F $R/gpurun_out/r02a/kernel_trace.csv
python $R/profiles/tools/timeline.py $F $R/gpurun_out/r02a/timeline.txt
cd $R
HF_POLL=0 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench_nopoll.json 2>/dev/null
HF_HOST_TRACE=1 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench_trace.json 2> gpurun_out/r02a/host_trace.err
cat gpurun_out/r02a/bench.json | cut -c1-400
tail -3 gpurun_out/r02a/host_trace.err
rocminfo | grep -E "Compute Unit|Max Clock|Marketing" | head; nproc; ls /opt/rocm/include/rccl/ | head
