#!/bin/bash
# Tier 1 of oracle/ref_recipe: builds the three harnesses against a real Eigen 3.3 + OpenCV 3.x and the reference's own
# kabschEst.cpp, runs them, collects tests/golden/ref_*.npz.  Refuses to do anything when the libraries are missing
# (no stand-in headers: the point is the real ones).  REF = reference checkout (default /root/reference).
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd); REF=${REF:-/root/reference}; OUT=$ROOT/oracle/_ref
EIGEN_INC=${EIGEN_INC:-$(for d in /usr/include/eigen3 /usr/local/include/eigen3; do if [ -f $d/Eigen/Core ]; then echo $d; break; fi; done)}
if [ -z "$EIGEN_INC" ]; then echo "ref_recipe: Eigen 3.3 headers not found (set EIGEN_INC): parity stays unpinned" >&2; exit 3; fi
if ! pkg-config --exists opencv 2>/dev/null && ! pkg-config --exists opencv4 2>/dev/null && [ -z "$OPENCV_FLAGS" ]; then
  echo "ref_recipe: OpenCV not found (set OPENCV_FLAGS='-I... -L... -lopencv_core -lopencv_features2d'): parity stays unpinned" >&2; exit 3; fi
CVF=${OPENCV_FLAGS:-$(pkg-config --cflags --libs opencv 2>/dev/null || pkg-config --cflags --libs opencv4)}
FLAGS="-O2 -msse2 -DEIGEN_DONT_VECTORIZE -std=c++11 -I$EIGEN_INC -I$REF/include/putslam -I$REF/include -I$REF"
mkdir -p $OUT
python3 $HERE/make_inputs.py
g++ $FLAGS $HERE/eigen_core_harness.cpp -o $OUT/eigen_core_harness
g++ $FLAGS $HERE/kabsch_harness.cpp $REF/src/TransformEst/kabschEst.cpp -o $OUT/kabsch_harness $CVF
g++ $FLAGS $HERE/bfmatcher_harness.cpp -o $OUT/bfmatcher_harness $CVF
$OUT/eigen_core_harness $OUT/inputs/eigen_core.bin $OUT/eigen_core.out
$OUT/kabsch_harness $OUT/inputs/kabsch.bin $OUT/kabsch.out
$OUT/bfmatcher_harness $OUT/inputs/pairs.bin $OUT/bfmatcher.out
python3 $HERE/collect.py
