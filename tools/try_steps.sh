#!/bin/bash
for f in build_variants/lib_*.so; do cp "$f" linux-fg_amd/liblinuxfg_hip.so; echo "== $f"; python tools/scale_steps.py | tail -4; done
