// On-device data side: the leap-frog simulators that generate the reference's datasets.
//   ElectrostaticFieldSim.sample_trajectory   experiments/electrostatic/dataset/electrostatic_field_sim.py:63-170
//   GravitationalFieldSim.sample_trajectory   experiments/gravitational/dataset/gravitational_field_sim.py:75-131
// The reference integrates one simulation at a time in numpy (fp64, T = 5000 sequential steps of an
// [M, M] pair computation, 70 000 simulations per dataset).  Here a simulation is a group of M lanes
// (one per ball, moving balls first, then the static field sources); 64 / M simulations share a wavefront,
// one wavefront per workgroup; positions are exchanged through LDS every step; each lane adds the pair terms
// in the reference's order (j = 0 .. M-1) with every product and sum rounded separately (no FMA contraction),
// so the trajectories follow numpy's to rounding of pow / sqrt.  fp64 VALU-bound: nothing here touches HBM
// except the T / sample_freq saved frames.
#pragma once
#include "common.h"

namespace {

constexpr int SIM_MAX_BALLS = 64;

// loc0, vel0 [S][M][D], charges [S][M] (static charges already scaled); loc, vel [S][T_save][M][D];
// maxed [S] (zeroed by the host): how often a force was capped (printed by the reference, :168).
template <int D>
__global__ void __launch_bounds__(64)
k_sim_electrostatic(const double* __restrict__ loc0, const double* __restrict__ vel0, const double* __restrict__ charges,
                    int64_t n_sims, int n_balls, int M, int T, int sample_freq, double strength, double dt,
                    double max_F, double* __restrict__ loc, double* __restrict__ vel, int64_t* __restrict__ maxed) {
#pragma clang fp contract(off)
    __shared__ double pos[64 * D];
    __shared__ double qs[64];
    const int lane = threadIdx.x;
    const int spw = 64 / M;                                      // simulations per wavefront
    const int sl = lane / M, i = lane - sl * M;
    const int64_t s = (int64_t)blockIdx.x * spw + sl;
    const bool active = sl < spw && s < n_sims;
    const bool moving = active && i < n_balls;
    const int T_save = T / sample_freq - 1;
    double x[D], v[D], q = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) { x[d] = 0.0; v[d] = 0.0; }
    if (active) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            x[d] = loc0[((size_t)s * M + i) * D + d];
            v[d] = vel0[((size_t)s * M + i) * D + d];
        }
        q = charges[(size_t)s * M + i];
        // frames of the static balls never change (:105); frame 0 of the moving ones is overwritten at the first save
        for (int c = 0; c < T_save; ++c)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                loc[(((size_t)s * T_save + c) * M + i) * D + d] = (c == 0 || i >= n_balls) ? x[d] : 0.0;
                vel[(((size_t)s * T_save + c) * M + i) * D + d] = c == 0 ? v[d] : 0.0;
            }
    }
    double* mine = pos + (size_t)sl * M * D;                     // this simulation's positions
    qs[lane] = q;
    const double* myq = qs + sl * M;
    int64_t capped = 0;
    int counter = 0;
    for (int step = 0; step < T; ++step) {
        if (step > 0) {
            if (moving) {
#pragma unroll
                for (int d = 0; d < D; ++d) x[d] = x[d] + dt * v[d];                    // :137
            }
            if (step % sample_freq == 0) {                                                // :139-142
                if (moving) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        loc[(((size_t)s * T_save + counter) * M + i) * D + d] = x[d];
                        vel[(((size_t)s * T_save + counter) * M + i) * D + d] = v[d];
                    }
                }
                ++counter;
            }
        }
        if (active) {
#pragma unroll
            for (int d = 0; d < D; ++d) mine[i * D + d] = x[d];
        }
        __syncthreads();
        if (active) {
            double F[D];
#pragma unroll
            for (int d = 0; d < D; ++d) F[d] = 0.0;
            for (int j = 0; j < M; ++j) {
                double diff[D], l2 = 0.0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    diff[d] = x[d] - mine[j * D + d];
                    l2 = l2 + diff[d] * diff[d];                                          // cdist 'sqeuclidean' (:45)
                }
                const double fs = j == i ? 0.0 : (strength * (q * myq[j])) / (l2 * sqrt(l2)); // :144-146
#pragma unroll
                for (int d = 0; d < D; ++d) F[d] = F[d] + fs * diff[d];                   // :147-149
            }
            double n2 = 0.0;
#pragma unroll
            for (int d = 0; d < D; ++d) n2 = n2 + F[d] * F[d];
            const double norm = sqrt(n2);
            if (norm > max_F) {                                                           // :152-157
#pragma unroll
                for (int d = 0; d < D; ++d) F[d] = (max_F * F[d]) / norm;
                ++capped;
            }
            if (moving) {
#pragma unroll
                for (int d = 0; d < D; ++d) v[d] = v[d] + dt * F[d];                      // :161
            }
        }
        __syncthreads();
    }
    if (active && capped != 0) atomicAdd(reinterpret_cast<unsigned long long*>(maxed + s), (unsigned long long)capped);
}

// pos0, vel0 [S][M][D] (velocities already in the centre-of-mass frame), mass [S][M];
// pos, vel, force [S][T_save][M][D], T_save = T / sample_freq.  Kick-drift-kick with Plummer softening.
template <int D>
__global__ void __launch_bounds__(64)
k_sim_gravitational(const double* __restrict__ pos0, const double* __restrict__ vel0, const double* __restrict__ mass,
                    int64_t n_sims, int n_balls, int M, int T, int sample_freq, double G, double dt, double softening,
                    double* __restrict__ pos_save, double* __restrict__ vel_save, double* __restrict__ force_save) {
#pragma clang fp contract(off)
    __shared__ double pos[64 * D];
    __shared__ double ms[64];
    const int lane = threadIdx.x;
    const int spw = 64 / M;
    const int sl = lane / M, i = lane - sl * M;
    const int64_t s = (int64_t)blockIdx.x * spw + sl;
    const bool active = sl < spw && s < n_sims;
    const bool moving = active && i < n_balls;
    const int T_save = T / sample_freq;
    const double soft2 = softening * softening;
    double x[D], v[D], a[D], m = 0.0;
#pragma unroll
    for (int d = 0; d < D; ++d) { x[d] = 0.0; v[d] = 0.0; a[d] = 0.0; }
    if (active) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            x[d] = pos0[((size_t)s * M + i) * D + d];
            v[d] = vel0[((size_t)s * M + i) * D + d];
        }
        m = mass[(size_t)s * M + i];
    }
    double* mine = pos + (size_t)sl * M * D;
    ms[lane] = m;
    const double* mym = ms + sl * M;
    auto accelerate = [&]() {                                    // compute_acceleration, :34-43
        if (active) {
#pragma unroll
            for (int d = 0; d < D; ++d) mine[i * D + d] = x[d];
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int d = 0; d < D; ++d) a[d] = 0.0;
            for (int j = 0; j < M; ++j) {
                double diff[D], r2 = 0.0;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    diff[d] = mine[j * D + d] - x[d];
                    r2 = r2 + diff[d] * diff[d];
                }
                r2 = r2 + soft2;
                const double inv_r3 = r2 > 0.0 ? 1.0 / (r2 * sqrt(r2)) : r2;
#pragma unroll
                for (int d = 0; d < D; ++d) a[d] = a[d] + (G * (diff[d] * inv_r3)) * mym[j];
            }
        }
        __syncthreads();
    };
    accelerate();                                                // :99
    for (int step = 0; step < T; ++step) {
        if (step % sample_freq == 0 && active) {                 // :102-110
            const size_t o = (((size_t)s * T_save + step / sample_freq) * M + i) * D;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                pos_save[o + d] = x[d];
                vel_save[o + d] = step == 0 ? 0.0 : v[d];
                force_save[o + d] = step == 0 ? 0.0 : a[d] * m;
            }
        }
        if (moving) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                v[d] = v[d] + (a[d] * dt) / 2.0;                 // :113
                x[d] = x[d] + v[d] * dt;                         // :116
            }
        }
        accelerate();                                            // :119
        if (moving) {
#pragma unroll
            for (int d = 0; d < D; ++d) v[d] = v[d] + (a[d] * dt) / 2.0;   // :122
        }
    }
}

// The n-body simulators behind the state2state runner's data sets (experiments/lorentz/dataset/synthetic_sim.py):
//   ChargedParticlesSim.sample_trajectory :221-300   ('charged':  Coulomb forces only)
//   GravitySim.sample_trajectory          :375-460   ('static':   + a constant force 0.098 on z)
//   DynamicSim.sample_trajectory          :536-622   ('dynamic':  + the Lorentz force q (v x B), B = 0.5 (1, 1, 1))
// 3-D, all balls move, squared distances from the expansion |a|^2 + |b|^2 - 2 a.b + 1e-6 (:167-178), every force
// COMPONENT clipped to +-max_F (:276-277), leap-frog; frames are stored [T_save][3][n] as the reference does.
//   FixCharge.sample_trajectory           :700-790   ('fixcharge': + the Coulomb force of a fixed charge at ext)
//   SpringSim.sample_trajectory           :74-146    ('springs':  pair forces -k edges_ij (x_i - x_j), no charges)
// ext_mode 0: none, 1: F += ext (constant vector), 2: F += q (v x ext), 3: F += ext_strength q (x - ext) / |x - ext|^3.
// pair != NULL: the pair force is -strength * pair[s][i][j] (x_i - x_j) instead of Coulomb's (springs).
__global__ void __launch_bounds__(64)
k_sim_charged(const double* __restrict__ loc0, const double* __restrict__ vel0, const double* __restrict__ charges,
              const double* __restrict__ pair, int64_t n_sims, int M, int T, int sample_freq, double strength, double dt,
              double max_F, int ext_mode, double e0, double e1, double e2, double ext_strength,
              double* __restrict__ loc, double* __restrict__ vel) {
#pragma clang fp contract(off)
    constexpr int D = 3;
    __shared__ double pos[64 * D];
    __shared__ double nrm[64];
    __shared__ double qs[64];
    const int lane = threadIdx.x;
    const int spw = 64 / M;
    const int sl = lane / M, i = lane - sl * M;
    const int64_t s = (int64_t)blockIdx.x * spw + sl;
    const bool active = sl < spw && s < n_sims;
    const int T_save = T / sample_freq - 1;
    double x[D] = {0.0, 0.0, 0.0}, v[D] = {0.0, 0.0, 0.0}, q = 0.0;
    if (active) {
#pragma unroll
        for (int d = 0; d < D; ++d) {                                  // inputs [S][3][M], as the reference lays them out
            x[d] = loc0[((size_t)s * D + d) * M + i];
            v[d] = vel0[((size_t)s * D + d) * M + i];
        }
        q = charges != nullptr ? charges[(size_t)s * M + i] : 0.0;
        if (T_save > 0) {
#pragma unroll
            for (int d = 0; d < D; ++d) {                              // frame 0 = the initial state until the first save
                loc[(((size_t)s * T_save) * D + d) * M + i] = x[d];
                vel[(((size_t)s * T_save) * D + d) * M + i] = v[d];
            }
        }
    }
    double* mine = pos + (size_t)sl * M * D;
    double* myn = nrm + sl * M;
    qs[lane] = q;
    const double* myq = qs + sl * M;
    int counter = 0;
    for (int step = 0; step < T; ++step) {
        if (step > 0) {
            if (active) {
#pragma unroll
                for (int d = 0; d < D; ++d) x[d] = x[d] + dt * v[d];
            }
            if (step % sample_freq == 0) {
                if (active) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        loc[(((size_t)s * T_save + counter) * D + d) * M + i] = x[d];
                        vel[(((size_t)s * T_save + counter) * D + d) * M + i] = v[d];
                    }
                }
                ++counter;
            }
        }
        if (active) {
#pragma unroll
            for (int d = 0; d < D; ++d) mine[i * D + d] = x[d];
            myn[i] = (x[0] * x[0] + x[1] * x[1]) + x[2] * x[2];
        }
        __syncthreads();
        if (active) {
            double F[D] = {0.0, 0.0, 0.0};
            const double ni = myn[i];
            if (pair != nullptr) {                                                      // springs, :98-110
                const double* prow = pair + ((size_t)s * M + i) * M;
                for (int j = 0; j < M; ++j) {
                    const double fs = j == i ? 0.0 : -strength * prow[j];
#pragma unroll
                    for (int d = 0; d < D; ++d) F[d] = F[d] + fs * (x[d] - mine[j * D + d]);
                }
            } else {
                for (int j = 0; j < M; ++j) {
                    const double dot = (x[0] * mine[j * D] + x[1] * mine[j * D + 1]) + x[2] * mine[j * D + 2];
                    const double l2 = ((ni + myn[j]) - 2.0 * dot) + 1e-6;               // _l2, :167-178
                    const double fs = j == i ? 0.0 : (strength * (q * myq[j])) / (l2 * sqrt(l2));
#pragma unroll
                    for (int d = 0; d < D; ++d) F[d] = F[d] + fs * (x[d] - mine[j * D + d]);
                }
            }
            if (ext_mode == 1) {
                F[0] = F[0] + e0; F[1] = F[1] + e1; F[2] = F[2] + e2;
            } else if (ext_mode == 2) {                                                  // np.cross(v, B) * q
                F[0] = F[0] + (v[1] * e2 - v[2] * e1) * q;
                F[1] = F[1] + (v[2] * e0 - v[0] * e2) * q;
                F[2] = F[2] + (v[0] * e1 - v[1] * e0) * q;
            } else if (ext_mode == 3) {                                                  // FixCharge, :742-746
                const double d0 = x[0] - e0, d1 = x[1] - e1, d2 = x[2] - e2;
                const double r2 = (d0 * d0 + d1 * d1) + d2 * d2;
                const double c = (ext_strength * q) / (r2 * sqrt(r2));
                F[0] = F[0] + c * d0; F[1] = F[1] + c * d1; F[2] = F[2] + c * d2;
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                F[d] = F[d] > max_F ? max_F : F[d];
                F[d] = F[d] < -max_F ? -max_F : F[d];
                v[d] = v[d] + dt * F[d];
            }
        }
        __syncthreads();
    }
}

}  // namespace
