// usac_harness.cpp -- TEST INFRASTRUCTURE (oracle/): drives the REFERENCE's own USAC<ProblemType> (include/putslam/USAC/USAC.h,
// compiled where it lies under /root/reference by oracle/ref_usac/build.sh; nothing of it is copied) so that the oracle's
// restatement of SURVEY 8a row A11 can be checked against the reference's code instead of against a reading of it:
//   stop     USAC<T>::updateStandardStopping (USAC.h:944-971) for (numInliers, totPoints) pairs
//   solve    USAC<T>::solve (USAC.h:296-520) under RANSAC_USAC's configuration (USAC_wrapper.cpp:62-100: confidence 0.99, sample
//            size 3, 850 000 hypotheses at most, uniform sampling, standard verification, no local optimisation) over a REPLAYED
//            sequence of per-hypothesis outcomes (model valid or not, inlier count): iterations made, best count, the hypothesis
//            that was stored last
//   sample   USAC<T>::generateUniformRandomSample (USAC.h:562-579) with rand() supplied by this file from the oracle's counter
//            stream po_draw31(seed, hypothesis, draw): the 3-point samples the reference's rule makes of those draws
// Input (stdin, text):   "stop N" + N lines "inliers total" | "solve M H" + H lines "valid count" | "sample SEED M H"
// Output (stdout, text): one line per query (see main).  The reference's solve() prints a line per iteration to std::cout: the
// stream is disabled while it runs and the results go to stdout through printf.
#include <math.h> // (USAC.h calls log / exp / ceil unqualified and leaves the header to its includer, as PUTSLAMEstimator.h does)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <iostream>
#include <string>
#include <vector>

#include "USAC.h" // the reference's header (found through -I /root/reference/include/putslam/USAC, see build.sh)

extern "C" uint32_t po_draw31(uint64_t seed, uint32_t h, uint32_t j); // oracle/putslam_oracle.c: the build's sample stream

// ---- rand() for the reference's sampler: the oracle's counter stream of the current hypothesis
static uint64_t g_seed = 0;
static uint32_t g_hyp = 0, g_draw = 0;
static bool g_own_rand = false;
extern "C" int rand(void) noexcept
{
    // (outside "sample" / "solve": std::random_shuffle of initDataUSAC's evaluation pool, unused by standard verification)
    if (!g_own_rand) return (int)po_draw31(0x5EEDull, 0xFFFFFFu, g_draw++);
    return (int)po_draw31(g_seed, g_hyp, g_draw++);
}

class Replay : public USAC<Replay>
{
public:
    std::vector<int> valid, count;
    std::vector<std::vector<unsigned int>> samples;
    long stored = -1;
    bool record_samples = false;

    unsigned int generateMinimalSampleModels()
    {
        const unsigned h = usac_results_.hyp_count_ - 1; // (solve() has already counted this iteration)
        if (record_samples) samples.push_back(min_sample_);
        g_hyp = h + 1; // the NEXT iteration's draws
        g_draw = 0;
        return (h < valid.size() && valid[h]) ? 1u : 0u;
    }
    bool generateRefinedModel(std::vector<unsigned int> &, const unsigned int, bool = false, double * = NULL) { return true; } // (defaults as in PUTSLAMEstimator.h)
    bool validateSample() { return true; }
    bool validateModel(unsigned int) { return true; }
    bool evaluateModel(unsigned int, unsigned int *numInliers, unsigned int *numPointsTested)
    {
        const unsigned h = usac_results_.hyp_count_ - 1;
        *numInliers = h < count.size() ? (unsigned)count[h] : 0u;
        *numPointsTested = usac_num_data_points_;
        return true;
    }
    void testSolutionDegeneracy(bool *d, bool *u) { *d = false; *u = false; }
    unsigned int upgradeDegenerateModel() { return 0; }
    void findWeights(unsigned int, const std::vector<unsigned int> &, unsigned int, double *) {}
    void storeModel(unsigned int, unsigned int) { stored = (long)usac_results_.hyp_count_ - 1; }

    unsigned int stop(unsigned a, unsigned b, unsigned c) { return updateStandardStopping(a, b, c); }
    void draw_sample(unsigned M, std::vector<unsigned int> *s) { generateUniformRandomSample(M, 3, s); }
};

static ConfigParams wrapper_config(unsigned M) // RANSAC_USAC::init_usac_configuration, USAC_wrapper.cpp:62-100
{
    ConfigParams cfg;
    cfg.common.confThreshold = 0.99;
    cfg.common.minSampleSize = 3;
    cfg.common.inlierThreshold = 0.02;
    cfg.common.maxHypotheses = 850000;
    cfg.common.maxSolutionsPerSample = 1;
    cfg.common.prevalidateSample = false;
    cfg.common.prevalidateModel = false;
    cfg.common.numDataPoints = M; // (USAC_wrapper.cpp:183: the number of matches of the call)
    cfg.common.testDegeneracy = false;
    cfg.common.randomSamplingMethod = USACConfig::SAMP_UNIFORM;
    cfg.common.verifMethod = USACConfig::VERIF_STANDARD;
    cfg.common.localOptMethod = USACConfig::LO_NONE;
    return cfg;
}

int main()
{
    char word[32];
    while (std::scanf("%31s", word) == 1) {
        const std::string w(word);
        if (w == "stop") {
            int n = 0;
            if (std::scanf("%d", &n) != 1) return 2;
            Replay r;
            r.initParamsUSAC(wrapper_config(10));
            for (int i = 0; i < n; ++i) {
                unsigned a, b;
                if (std::scanf("%u %u", &a, &b) != 2) return 2;
                std::printf("%u\n", r.stop(a, b, 3));
            }
        } else if (w == "solve") {
            unsigned M = 0;
            int H = 0;
            if (std::scanf("%u %d", &M, &H) != 2) return 2;
            Replay r;
            r.valid.resize((size_t)H);
            r.count.resize((size_t)H);
            for (int i = 0; i < H; ++i)
                if (std::scanf("%d %d", &r.valid[(size_t)i], &r.count[(size_t)i]) != 2) return 2;
            const ConfigParams cfg = wrapper_config(M);
            r.initParamsUSAC(cfg);
            r.initDataUSAC(cfg);
            g_seed = 0xC0FFEEull; // (the samples solve() draws decide nothing here: the outcomes are replayed)
            g_hyp = 0;
            g_draw = 0;
            g_own_rand = true;
            std::cout.setstate(std::ios_base::failbit); // (solve() narrates every iteration)
            const bool ok = r.solve();
            std::cout.clear();
            g_own_rand = false;
            std::printf("%d %u %u %ld\n", ok ? 1 : 0, r.usac_results_.hyp_count_, r.usac_results_.best_inlier_count_, r.stored);
        } else if (w == "sample") {
            unsigned long long seed = 0;
            unsigned M = 0;
            int H = 0;
            if (std::scanf("%llu %u %d", &seed, &M, &H) != 3) return 2;
            Replay r;
            r.initParamsUSAC(wrapper_config(M));
            g_seed = seed;
            g_own_rand = true;
            for (int h = 0; h < H; ++h) {
                std::vector<unsigned int> s(3, 0xFFFFFFFFu);
                g_hyp = (uint32_t)h;
                g_draw = 0;
                r.draw_sample(M, &s);
                std::printf("%u %u %u\n", s[0], s[1], s[2]);
            }
            g_own_rand = false;
        } else {
            return 2;
        }
    }
    return 0;
}
