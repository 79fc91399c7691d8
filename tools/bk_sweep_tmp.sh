for st in 4 3 2 1 4; do echo "== kc streams $st"; GMSX_KC_STREAMS=$st python tools/kc_probe.py 22 24 --k 4 2>&1 | tail -2 | cut -c1-200; done
