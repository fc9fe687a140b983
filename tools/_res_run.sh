set -e
for d in 20 24 28 32 36; do echo "delay $d"; SPRINGCRAFT_RESIDENT_DELAY=$d timeout -k 10 300 python tools/resident_check.py 128 300 900 1536 2048 3000 2>&1 | grep "per-column" | cut -c1-70; done
SPRINGCRAFT_RESIDENT_DELAY=28 timeout -k 10 300 python tools/resident_check.py --phases 300 1536 2048 3000 2>&1 | grep "n="
