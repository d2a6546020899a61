for c in 2; do echo "== config $c"; BENCH_ARGS="--config $c" bash profiles/tools/r03_ab.sh "-DHF_DUMMY" "-DHF_PS_PIPE" "-DHF_DUMMY" "-DHF_PS_PIPE"; done
