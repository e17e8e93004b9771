// print_defaults.cpp -- the drop-in's compiled-in defaults (FrameMatcher::MatcherParameters, the values the reference reads
// from resources/putslammatcherOpenCVParameters.xml and resources/datasetConfig/freiburg1_desk.xml) as one JSON line, for
// tests/test_reference_defaults.py.  Touches no GPU: the struct's constructor only fills numbers.
#include <cstdio>

#include "putslam_dropin.h"

int main()
{
    putslam_hip::FrameMatcher::MatcherParameters p;
    const RANSAC::parameters &r = p.RANSACParams;
    std::printf("{\"verbose\": %d, \"errorVersionVO\": %d, \"errorVersionMap\": %d, \"inlierThresholdEuclidean\": %.17g, "
                "\"inlierThresholdReprojection\": %.17g, \"inlierThresholdMahalanobis\": %.17g, \"minimalInlierRatioThreshold\": %.17g, "
                "\"minimalNumberOfMatches\": %d, \"usedPairs\": %d, \"matchingXYZSphereRadius\": %.17g, "
                "\"matchingXYZacceptRatioOfBestMatch\": %.17g, \"K\": [",
                r.verbose, r.errorVersionVO, r.errorVersionMap, r.inlierThresholdEuclidean, r.inlierThresholdReprojection,
                r.inlierThresholdMahalanobis, r.minimalInlierRatioThreshold, r.minimalNumberOfMatches, r.usedPairs,
                p.OpenCVParams.matchingXYZSphereRadius, p.OpenCVParams.matchingXYZacceptRatioOfBestMatch);
    for (int i = 0; i < 9; ++i) std::printf("%s%.9g", i ? ", " : "", (double)p.cameraMatrixMat.at<float>(i / 3, i % 3));
    std::printf("]}\n");
    return 0;
}
