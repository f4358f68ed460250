#!/bin/sh
# Build, then run the tunnel with the reference's defaults (128x64x64, 100 steps); any arguments are
# passed on to simulation.out (--grid, --steps, --stl ..., see src/main.cpp).  Frames land in ./data,
# where the reference's viewer picks them up:  python /path/to/reference/GUI/main.py
set -e
mkdir -p data
make
exec ./simulation.out "$@"
