#!/bin/bash
for L in "$@"; do echo "== $L"; SPRINGCRAFT_HIP_LIB=$L python tools/ab_small.py 2>/dev/null; done
