for c in 2 4; do echo "== config $c"; BENCH_ARGS="--config $c --steps 400 --warmup 100" bash profiles/tools/r03_ab.sh "-DHF_SEG_NO_REORDER" "-DHF_DUMMY" "-DHF_SEG_NO_REORDER" "-DHF_DUMMY"; done
