for t in 64 128 256 512; do echo "== RAMX_CP_DEV_THREADS=$t"; RAMX_CP_DEV_THREADS=$t timeout -k 10 200 python tools/cp_dev_timing.py 250 1000 2000 4000 2>&1 | grep "^W"; done
