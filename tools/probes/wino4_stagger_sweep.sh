for s in 0 1 2 3 4 6 8; do echo "stagger $s"; SCIPNP_W4_STAGGER=$s python tools/probes/wino4_check.py 2>&1 | grep "FFDNet body\|64->64\|128->128"; done
