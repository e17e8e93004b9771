#!/bin/bash
# Tier 1 of oracle/ref_recipe: builds the three harnesses against a real Eigen 3.3 + OpenCV 3.x and the reference's own
# kabschEst.cpp, runs them, collects tests/golden/ref_*.npz.  Refuses to do anything when the libraries are missing
# (no stand-in headers: the point is the real ones).  REF = reference checkout (default /root/reference).
#
#   run.sh          the real thing
#   run.sh --dry    resolves every path and prints every command it WOULD run, builds and runs nothing; exit code 0 when every
#                   file of the recipe (and, where $REF exists, of the reference) is where the commands expect it -- missing
#                   libraries are reported, not fatal: tests/test_ref_recipe_plumbing.py calls this so the recipe cannot rot
#                   in an image that has neither Eigen nor OpenCV.
set -e
DRY=0; if [ "${1:-}" = "--dry" ]; then DRY=1; fi
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd); REF=${REF:-/root/reference}; OUT=$ROOT/oracle/_ref
EIGEN_INC=${EIGEN_INC:-$(for d in /usr/include/eigen3 /usr/local/include/eigen3; do if [ -f $d/Eigen/Core ]; then echo $d; break; fi; done)}
HAVE_CV=1
if ! pkg-config --exists opencv 2>/dev/null && ! pkg-config --exists opencv4 2>/dev/null && [ -z "${OPENCV_FLAGS:-}" ]; then HAVE_CV=0; fi
if [ $DRY = 0 ]; then
  if [ -z "$EIGEN_INC" ]; then echo "ref_recipe: Eigen 3.3 headers not found (set EIGEN_INC): parity stays unpinned" >&2; exit 3; fi
  if [ $HAVE_CV = 0 ]; then
    echo "ref_recipe: OpenCV not found (set OPENCV_FLAGS='-I... -L... -lopencv_core -lopencv_features2d'): parity stays unpinned" >&2; exit 3; fi
fi
if [ $HAVE_CV = 1 ]; then CVF=${OPENCV_FLAGS:-$(pkg-config --cflags --libs opencv 2>/dev/null || pkg-config --cflags --libs opencv4)}; else CVF="<OPENCV_FLAGS: not found here>"; fi
EI=${EIGEN_INC:-"<EIGEN_INC: not found here>"}
FLAGS="-O2 -msse2 -DEIGEN_DONT_VECTORIZE -std=c++11 -I$EI -I$REF/include/putslam -I$REF/include -I$REF"
MISSING=0
need() { if [ ! -e "$1" ]; then echo "ref_recipe: MISSING $1" >&2; MISSING=1; fi; }
run() { if [ $DRY = 1 ]; then echo "+ $*"; else "$@"; fi; }
for f in make_inputs.py collect.py eigen_core_harness.cpp kabsch_harness.cpp bfmatcher_harness.cpp; do need $HERE/$f; done
if [ -d "$REF" ]; then
  # what the harnesses include / compile from the reference, where it lies
  need $REF/src/TransformEst/kabschEst.cpp
  need $REF/include/putslam/TransformEst/kabschEst.h
  need $REF/include/putslam/TransformEst/transformEst.h
  need $REF/include/putslam/Defs/putslam_defs.h
elif [ $DRY = 1 ]; then echo "ref_recipe: no reference checkout at $REF (its files are not checked)" >&2
fi
if [ $DRY = 1 ]; then
  echo "ref_recipe dry run: Eigen ${EIGEN_INC:-NOT FOUND}, OpenCV $([ $HAVE_CV = 1 ] && echo found || echo NOT FOUND), reference $([ -d "$REF" ] && echo $REF || echo NOT FOUND)"
fi
run mkdir -p $OUT
run python3 $HERE/make_inputs.py
run g++ $FLAGS $HERE/eigen_core_harness.cpp -o $OUT/eigen_core_harness
run g++ $FLAGS $HERE/kabsch_harness.cpp $REF/src/TransformEst/kabschEst.cpp -o $OUT/kabsch_harness $CVF
run g++ $FLAGS $HERE/bfmatcher_harness.cpp -o $OUT/bfmatcher_harness $CVF
run $OUT/eigen_core_harness $OUT/inputs/eigen_core.bin $OUT/eigen_core.out
run $OUT/kabsch_harness $OUT/inputs/kabsch.bin $OUT/kabsch.out
run $OUT/bfmatcher_harness $OUT/inputs/pairs.bin $OUT/bfmatcher.out
run python3 $HERE/collect.py
if [ $DRY = 1 ]; then exit $MISSING; fi
