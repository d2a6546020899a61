R=$(pwd); export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/kt5 && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5 -o kt -- python $R/bench.py --config ${1:-5} --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1 )
F=$(find /tmp/kt5 -name '*kernel_trace.csv' | head -1)
python profiles/tools/timeline.py $F /dev/stdout
