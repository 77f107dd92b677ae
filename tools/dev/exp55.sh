timeout -k 10 600 python tools/dev/ab.py build_exp/final3.so build_exp/drainevery.so
