// kabsch_harness.cpp -- the reference's own KabschEst (src/TransformEst/kabschEst.cpp:24-68, compiled from the
// reference checkout by run.sh) on recorded point sets.  Never built in the development image (no Eigen / OpenCV).
#include "TransformEst/kabschEst.h"

#include <cstdint>
#include <cstdio>
#include <vector>

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = std::fopen(argv[1], "rb"), *o = std::fopen(argv[2], "wb");
    if (!f || !o) return 2;
    int32_t cases;
    if (std::fread(&cases, 4, 1, f) != 1) return 2;
    std::fwrite(&cases, 4, 1, o);
    putslam::TransformEst *est = putslam::createKabschEstimator();
    for (int c = 0; c < cases; ++c) {
        int32_t m;
        if (std::fread(&m, 4, 1, f) != 1) return 2;
        Eigen::MatrixXd A(m, 3), B(m, 3); // column-major, like the files
        if (std::fread(A.data(), 8, (size_t)m * 3, f) != (size_t)m * 3 || std::fread(B.data(), 8, (size_t)m * 3, f) != (size_t)m * 3) return 2;
        putslam::Mat34 &T = est->computeTransformation(A, B);
        Eigen::Matrix4d M = T.matrix();
        std::fwrite(&m, 4, 1, o);
        std::fwrite(M.data(), 8, 16, o);
    }
    std::fclose(f);
    std::fclose(o);
    return 0;
}
