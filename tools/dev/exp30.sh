python tools/dev/ab.py build_exp/head.so build_exp/mtnts.so build_exp/mtntl.so
