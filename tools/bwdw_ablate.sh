for v in "" _abl1 _abl2 _abl3 _abl8; do echo "variant=$v"; MT_LIB_VARIANT=libmtseg_hip$v.so python tools/bench_conv.py --mode bwdw --cin 32 --cout 32 --reps 5 --lazy ${LAZY:-1} 2>&1 | tail -1; done
