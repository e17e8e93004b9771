// rand_interpose.cpp -- replaces libc rand() / srand() for the RANSAC harness: draws come from a recorded counter stream
// so that a run of the reference's RANSAC::estimateTransformation (which calls srand(time(0)) in its constructor and
// rand() % M in getRandomMatches, RANSAC.cpp:13,180-205) is reproducible and its draws are known.
#include <cstdint>
#include <vector>

static uint64_t g_seed = 1, g_count = 0;
std::vector<int> g_rand_log;

static uint64_t mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

extern "C" void ref_rand_reset(uint64_t seed)
{
    g_seed = seed;
    g_count = 0;
    g_rand_log.clear();
}
extern "C" void srand(unsigned) {} // the constructor's srand(time(0)) must not disturb the stream
extern "C" int rand(void)
{
    int v = (int)(mix64(g_seed ^ mix64(g_count++)) >> 33); // 31 bits, like RAND_MAX = 2^31 - 1
    g_rand_log.push_back(v);
    return v;
}
