for c in 2; do echo "== config $c"; BENCH_ARGS="--config $c" bash profiles/tools/r03_ab.sh "-DHF_DUMMY" "-DHF_SEG_STAGGER=30" "-DHF_SEG_STAGGER=60" "-DHF_SEG_STAGGER=120" "-DHF_DUMMY"; done
