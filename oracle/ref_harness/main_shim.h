// TEST INFRASTRUCTURE ONLY (oracle/_ref build).
//
// Pre-empts the reference's src/main.h through its include guard so that the
// UNMODIFIED reference headers (compiled where they lie under /root/reference)
// draw their random numbers from a re-seedable per-(pixel,sample) stream
// instead of the process-global std::mt19937 of src/main.h:28-32.
// Everything else src/main.h declares (src/main.h:12-26,34-49) is declared here
// with the same meaning, because the rest of the reference relies on it.
#ifndef MAIN_H
#define MAIN_H

#include <cmath>
#include <cstdlib>
#include <iostream>
#include <limits>
#include <memory>
#include <random>
#include <string>

#include "../oracle_rng.h"

using std::make_shared;
using std::shared_ptr;

const double infinity = std::numeric_limits<double>::infinity();
const double pi = 3.1415926535897932385;

inline double degrees_to_radians(double degrees) { return degrees * pi / 180.0; }

// --- the replaced generator -------------------------------------------------
struct harness_rng_state {
    int engine = 0;            // 0 = counter stream, 1 = serial mt19937 through std::uniform_real_distribution
    orng_stream stream{0, 0};
    std::mt19937 mt;           // default seeded (5489) like src/main.h:30
    std::uniform_real_distribution<double> dist{0.0, 1.0};
    unsigned long long total_draws = 0;
};
inline harness_rng_state& harness_rng() { static harness_rng_state s; return s; }

inline double random_double() {
    harness_rng_state& s = harness_rng();
    s.total_draws++;
    if (s.engine == 1) return s.dist(s.mt);
    return orng_stream_next(&s.stream);
}

inline double random_double(double min, double max) { return min + (max - min) * random_double(); }

inline int random_int(int min, int max) { return int(random_double(min, max + 1)); }

#include "color.h"
#include "interval.h"
#include "ray.h"
#include "vec3.h"

#endif
