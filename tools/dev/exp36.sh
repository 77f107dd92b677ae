timeout -k 10 600 python tools/dev/ab.py build_exp/head.so build_exp/head_hot.so build_exp/peel.so build_exp/peel_hot.so
