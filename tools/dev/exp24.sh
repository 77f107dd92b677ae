python tools/dev/ab.py build_exp/defer16.so build_exp/defer32.so
