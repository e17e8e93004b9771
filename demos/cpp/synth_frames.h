// synth_frames.h -- synthetic RGB-D feature frames for the C++ demos: a static cloud of keypoints with 256-bit descriptors seen
// from a camera on a smooth hand-held-style path, 2 mm point noise, 4 % descriptor bit noise, 12 % clutter, random keypoint
// order -- so that the estimated increments can be checked against the ground truth they were made from.  (Detection and
// description are image-domain stages outside the path: frames enter the demos where Matcher::match has descriptors and 3-D
// points, reference src/Matcher/matcher.cpp:467-480.)
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace synth {

struct Rng { // splitmix64
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed) {}
    uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double uni(double a, double b) { return a + (b - a) * uni(); }
    double gauss() { return std::sqrt(-2.0 * std::log(uni() + 1e-300)) * std::cos(6.283185307179586 * uni()); }
};

struct Pose { // camera-to-world
    double R[3][3], t[3];
};

// smooth path that stays near the start (the cloud must remain inside the 0.1-6 m depth window, RANSAC.cpp:65-74): about 2 cm
// and 0.3 degrees per frame
inline Pose camera_pose(int k)
{
    const double a = 0.12 * std::sin(0.045 * k), b = 0.15 * std::sin(0.035 * k), c = 0.10 * std::sin(0.05 * k);
    const double ca = std::cos(a), sa = std::sin(a), cb = std::cos(b), sb = std::sin(b), cc = std::cos(c), sc = std::sin(c);
    Pose P;
    const double Rz[3][3] = {{cc, -sc, 0}, {sc, cc, 0}, {0, 0, 1}}, Ry[3][3] = {{cb, 0, sb}, {0, 1, 0}, {-sb, 0, cb}},
                 Rx[3][3] = {{1, 0, 0}, {0, ca, -sa}, {0, sa, ca}};
    double T[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T[i][j] = 0;
            for (int m = 0; m < 3; ++m) T[i][j] += Ry[i][m] * Rx[m][j];
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            P.R[i][j] = 0;
            for (int m = 0; m < 3; ++m) P.R[i][j] += Rz[i][m] * T[m][j];
        }
    P.t[0] = 0.45 * std::sin(0.04 * k);
    P.t[1] = 0.12 * std::sin(0.1 * k);
    P.t[2] = 0.35 * std::sin(0.03 * k);
    return P;
}

// ground-truth increment, rows of the 3 x 4 matrix [R | t]: the current camera expressed in the previous one (what
// Matcher::match's estimatedTransformation approximates)
inline void increment(const Pose &prev, const Pose &cur, float G[12])
{
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            double v = 0;
            for (int m = 0; m < 3; ++m) v += prev.R[m][r] * cur.R[m][c];
            G[r * 4 + c] = (float)v;
        }
        double v = 0;
        for (int m = 0; m < 3; ++m) v += prev.R[m][r] * (cur.t[m] - prev.t[m]);
        G[r * 4 + 3] = (float)v;
    }
}

struct World {
    int N;
    Rng rng;
    std::vector<double> pos;    // N x 3, in front of the start pose
    std::vector<uint8_t> wdesc; // N x 32
    World(int n, uint64_t seed) : N(n), rng(seed), pos((size_t)n * 3), wdesc((size_t)n * 32)
    {
        for (int i = 0; i < N; ++i) {
            pos[3 * (size_t)i] = rng.uni(-2.5, 2.5);
            pos[3 * (size_t)i + 1] = rng.uni(-1.8, 1.8);
            pos[3 * (size_t)i + 2] = rng.uni(1.5, 5.0);
        }
        for (auto &b : wdesc) b = (uint8_t)rng.next();
    }
    // frame k: desc N x 32 bytes, pts N x 3 floats (12-byte stride)
    void observe(int k, uint8_t *desc, float *pts)
    {
        const Pose P = camera_pose(k);
        std::vector<int> order((size_t)N);
        for (int i = 0; i < N; ++i) order[(size_t)i] = i;
        for (int i = N - 1; i > 0; --i) std::swap(order[(size_t)i], order[(size_t)(rng.next() % (uint64_t)(i + 1))]);
        for (int s = 0; s < N; ++s) {
            const int i = order[(size_t)s];
            const double d[3] = {pos[3 * (size_t)i] - P.t[0], pos[3 * (size_t)i + 1] - P.t[1], pos[3 * (size_t)i + 2] - P.t[2]};
            double c[3];
            for (int r = 0; r < 3; ++r) c[r] = P.R[0][r] * d[0] + P.R[1][r] * d[1] + P.R[2][r] * d[2]; // R^T d
            const bool clutter = rng.uni() < 0.12;
            uint8_t *row = desc + (size_t)s * 32;
            float *p = pts + (size_t)s * 3;
            if (clutter) {
                for (int b = 0; b < 32; ++b) row[b] = (uint8_t)rng.next();
                p[0] = (float)rng.uni(-2, 2);
                p[1] = (float)rng.uni(-1.5, 1.5);
                p[2] = (float)rng.uni(0.5, 5.5);
            } else {
                std::memcpy(row, &wdesc[(size_t)i * 32], 32);
                for (int f = 0; f < 10; ++f) { // about 4 % of the bits flip between views
                    const unsigned bit = (unsigned)(rng.next() & 255u);
                    row[bit >> 3] ^= (uint8_t)(1u << (bit & 7u));
                }
                p[0] = (float)(c[0] + 0.002 * rng.gauss());
                p[1] = (float)(c[1] + 0.002 * rng.gauss());
                p[2] = (float)(c[2] + 0.002 * rng.gauss());
            }
        }
    }
};

} // namespace synth
