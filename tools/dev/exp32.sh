python tools/dev/ab.py build_exp/head.so build_exp/touch.so
