for k in 16 8 4 2; do echo "== RAMX_CP_K=$k"; RAMX_CP_K=$k timeout -k 10 200 python tools/cp_timing.py 2>&1 | grep -E "n  100|n  128|n   60|n  250"; done
