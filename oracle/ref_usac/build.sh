#!/bin/bash
# Builds oracle/_ref/usac_harness from the reference's own USAC.h + ConfigParams.h WHERE THEY LIE under /root/reference (g++ on
# the harness; the header is a template library, nothing else of the reference is needed: no OpenCV, no Eigen).  USAC.h includes
# "../include/USAC/ConfigParams.h" -- the layout of the stand-alone USAC distribution it was taken from -- so the include path
# gets a directory whose ../include/USAC is a symbolic link to the reference's include/putslam/USAC.  Outputs only under oracle/_ref/.
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(dirname "$(dirname "$HERE")")
REF=${PUTSLAM_REFERENCE:-/root/reference}
[ -f "$REF/include/putslam/USAC/USAC.h" ] || { echo "no reference tree at $REF: nothing built" >&2; exit 3; }
OUT=$ROOT/oracle/_ref
mkdir -p "$OUT/usac_inc/src" "$OUT/usac_inc/include"
ln -sfn "$REF/include/putslam/USAC" "$OUT/usac_inc/include/USAC"
make -s -C "$ROOT/oracle" >/dev/null
g++ -O2 -std=c++14 -w -I "$REF/include/putslam/USAC" -I "$OUT/usac_inc/src" "$HERE/usac_harness.cpp" -o "$OUT/usac_harness" \
    -L "$ROOT/oracle/_build" -lputslam_oracle -Wl,-rpath,"$ROOT/oracle/_build"
rm -f "$OUT/usac_inc/include/USAC"   # (a link into /root/reference must not travel)
echo "$OUT/usac_harness"
