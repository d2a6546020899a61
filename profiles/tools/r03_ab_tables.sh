for c in 2; do echo "== config $c"; BENCH_ARGS="--config $c" bash profiles/tools/r03_ab.sh "-DHF_SEG_PLAIN_DIV" "-DHF_DUMMY" "-DHF_SEG_PLAIN_DIV" "-DHF_DUMMY"; done
