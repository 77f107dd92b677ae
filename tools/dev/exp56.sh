SKYJO_LIB=$PWD/build_exp/stamps.so timeout -k 10 300 python tools/dev/stamps.py 65536 2>&1 | tail -12
