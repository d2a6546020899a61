for c in 2; do echo "== config $c"; BENCH_ARGS="--config $c" bash profiles/tools/r03_ab.sh "-DHF_DUMMY" "-DHF_SEG_LMAX=6 -DHF_SEG_OCC=4" "-DHF_SEG_LMAX=6 -DHF_SEG_OCC=3" "-DHF_DUMMY"; done
