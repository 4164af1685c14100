// input_grad.h -- gradients of the step with respect to its inputs x (positions) and vel (round 3; VERDICT r2 "missing 5").
// Included after backward.h.
//
// The reference's forward (nn/state2state/aether.py:169-186) is differentiable in x / vel.  aether_backward leaves behind
// everything this needs: DA = dL/d(layer-1 edge features) [E][FPAD] (sorted edge order), DN1 = dL/dn_1 [Nn][64] and
// DF = dL/df [Nn][16] (kb_field).  Positions and velocities enter through
//   * out = p + R(v) y                                   (aether.py:182-185; y = R^T (out - p) is recovered from the output)
//   * the field net z = [p | v | emb(q)]                  (aether.py:127-134: dz = W0^T dpre1, recomputed here from DF)
//   * rel_feat = [0 | R^T v | R^T f]                      (aether.py:39-48; through layer_1.res and the features' receiver columns)
//   * the edge features of j -> i in i's frame            (aether.py:52-92, geometry.py:7-101):
//       r = R_i^T (p_j - p_i), Euler angles of R_i^T R_j, |p_j - p_i|, bearing angle(s) of r, R_i^T v_j, R_i^T f_j.
// Everything that depends on a frame is collected as a matrix gradient GR = dL/dR (w = R^T u contributes u (x) dw; M = R_i^T R_j
// contributes R_j dM^T to R_i and R_i dM to R_j) and turned into angle gradients with dR/dtheta, dR/dphi at the end -- the
// same code for D = 2 and D = 3 -- and from there into dv through theta = atan2(v_y, v_x), phi = acos(clamp(v_z / (|v| + eps))).
// One 32-thread group per node walks the node's in-edges (as receiver) and out-edges (as sender); sums in fixed order.
#pragma once

namespace {

template <int D>
__device__ __forceinline__ void edge_local_grad(const float* __restrict__ da, const float (&rel)[D], const float* __restrict__ Rr,
                                                float (&dr)[D], float (&drel)[D]) {
    // receiver frame Rr (row-major R[b][a]); da = the edge's DA row.  dr = dL/dr (r = Rr^T rel) including the bearing
    // angles' share; drel = dL/d(rel) in global coordinates (through r and through |rel|).
    constexpr int O = D * (D - 1) / 2;
    constexpr int C_DIST = D + O, C_B = D + O + 1;
    float r[D];
#pragma unroll
    for (int a = 0; a < D; ++a) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < D; ++b) s += Rr[b * D + a] * rel[b];
        r[a] = s;
        dr[a] = da[a];
    }
    {   // bearing in the plane: atan2(r_y, r_x)  (symmetric theta, not normalised)
        const float den = r[0] * r[0] + r[1] * r[1];
        if (den > 0.f) {
            dr[0] += da[C_B] * (-r[1] / den);
            dr[1] += da[C_B] * (r[0] / den);
        }
    }
    if constexpr (D == 3) {   // polar angle acos(clamp(r_z / (|r| + eps), -1, 1))
        const float rho = sqrtf(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        const float c = r[2] / (rho + EPS_F);
        if (rho > 0.f && c > -1.0f && c < 1.0f) {
            const float dc = -da[C_B + 1] / sqrtf(1.0f - c * c);
            dr[2] += dc / (rho + EPS_F);
            const float k = -dc * r[2] / ((rho + EPS_F) * (rho + EPS_F) * rho);
#pragma unroll
            for (int a = 0; a < D; ++a) dr[a] += k * r[a];
        }
    }
    float d2 = 0.f;
#pragma unroll
    for (int b = 0; b < D; ++b) d2 += rel[b] * rel[b];
    const float dist = sqrtf(d2);
    const float kd = dist > 0.f ? da[C_DIST] / dist : 0.f;
#pragma unroll
    for (int b = 0; b < D; ++b) {
        float s = kd * rel[b];
#pragma unroll
        for (int a = 0; a < D; ++a) s += Rr[b * D + a] * dr[a];
        drel[b] = s;
    }
}

// dL/dM of the normalised Euler angles of M = R_i^T R_j (geometry.py:87-100; no clamp on the asin)
template <int D>
__device__ __forceinline__ void euler_grad(const float* __restrict__ da, const float* __restrict__ Ri, const float* __restrict__ Rj,
                                           float (&dM)[D][D]) {
    float M[D][D];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < D; ++b) s += Ri[b * D + a] * Rj[b * D + c];
            M[a][c] = s;
            dM[a][c] = 0.f;
        }
    const float ipi = 1.0f / PI_F;
    {
        const float den = M[0][0] * M[0][0] + M[1][0] * M[1][0];
        if (den > 0.f) {
            dM[0][0] += -M[1][0] / den * da[D] * ipi;
            dM[1][0] += M[0][0] / den * da[D] * ipi;
        }
    }
    if constexpr (D == 3) {
        const float s2 = 1.0f - M[2][0] * M[2][0];
        if (s2 > 0.f) dM[2][0] += -da[D + 1] * ipi / sqrtf(s2);
        const float den = M[2][1] * M[2][1] + M[2][2] * M[2][2];
        if (den > 0.f) {
            dM[2][1] += M[2][2] / den * da[D + 2] * ipi;
            dM[2][2] += -M[2][1] / den * da[D + 2] * ipi;
        }
    }
}

template <int D>
__global__ void __launch_bounds__(256)
kb_inputs(AetherParams P, const float* __restrict__ x, const float* __restrict__ vel, const float* __restrict__ charges,
          const float* __restrict__ nodeinfo, const float* __restrict__ out, const float* __restrict__ g_out,
          const float* __restrict__ DA, const float* __restrict__ DN1, const float* __restrict__ DF,
          const int32_t* __restrict__ rowptr, const int32_t* __restrict__ send_s, const int32_t* __restrict__ recv_s,
          const int32_t* __restrict__ srowptr, const int32_t* __restrict__ sperm, float* __restrict__ grad_x,
          float* __restrict__ grad_v, int64_t n_nodes,
          const float* __restrict__ field_gz = nullptr /* [n_nodes][2D]: dL/d[p | v] through an EXTERNAL field (the
          dynamic-field variant, aether_dynamic_field_backward_inputs); null = the built-in field net, recomputed below */,
          int hid = H /* width of DN1 / rows of W_res (wide.h) */) {
    using NI = NodeInfo<D>;
    constexpr int FIN = 2 * D + 16;
    constexpr int O = D * (D - 1) / 2;
    constexpr int C_RV = 2 * D + O, C_RF = 3 * D + O, C_CV = 5 * D + O, C_CF = 6 * D + O;
    __shared__ float sz[8][32], sh1[8][32], sd2[8][32], sd1[8][32];
    const int g = threadIdx.x >> 5, t = threadIdx.x & 31;
    const int64_t n = (int64_t)blockIdx.x * 8 + g;
    const bool ok = n < n_nodes;
    const int64_t nc = ok ? n : n_nodes - 1;
    const float* ni = nodeinfo + nc * NI::STRIDE;
    float gp[D], gv[D], GR[D][D], dcv[D], dcf[D];
#pragma unroll
    for (int b = 0; b < D; ++b) {
        gp[b] = 0.f; gv[b] = 0.f; dcv[b] = 0.f; dcf[b] = 0.f;
#pragma unroll
        for (int a = 0; a < D; ++a) GR[b][a] = 0.f;
    }
    // ---- as receiver i: in-edges j -> i, features in this node's frame
    for (int k = rowptr[nc] + t; k < rowptr[nc + 1]; k += 32) {
        const float* da = DA + (int64_t)k * FPAD;
        const float* nj = nodeinfo + (int64_t)send_s[k] * NI::STRIDE;
        float rel[D], dr[D], drel[D];
#pragma unroll
        for (int b = 0; b < D; ++b) rel[b] = nj[NI::P + b] - ni[NI::P + b];
        edge_local_grad<D>(da, rel, ni + NI::R, dr, drel);
        float dM[D][D];
        euler_grad<D>(da, ni + NI::R, nj + NI::R, dM);
#pragma unroll
        for (int b = 0; b < D; ++b) {
            gp[b] -= drel[b];
            dcv[b] += da[C_CV + b];
            dcf[b] += da[C_CF + b];
#pragma unroll
            for (int a = 0; a < D; ++a) {
                float s = rel[b] * dr[a] + nj[NI::V + b] * da[C_RV + a] + nj[NI::F + b] * da[C_RF + a];
#pragma unroll
                for (int c = 0; c < D; ++c) s += nj[NI::R + b * D + c] * dM[a][c];
                GR[b][a] += s;
            }
        }
    }
    // ---- as sender j: out-edges j -> i', features in the receiver's frame
    for (int kk = srowptr[nc] + t; kk < srowptr[nc + 1]; kk += 32) {
        const int64_t k = sperm[kk];
        const float* da = DA + k * FPAD;
        const float* nr = nodeinfo + (int64_t)recv_s[k] * NI::STRIDE;
        float rel[D], dr[D], drel[D];
#pragma unroll
        for (int b = 0; b < D; ++b) rel[b] = ni[NI::P + b] - nr[NI::P + b];
        edge_local_grad<D>(da, rel, nr + NI::R, dr, drel);
        float dM[D][D];
        euler_grad<D>(da, nr + NI::R, ni + NI::R, dM);
#pragma unroll
        for (int b = 0; b < D; ++b) {
            gp[b] += drel[b];
            float sv = 0.f;
#pragma unroll
            for (int a = 0; a < D; ++a) sv += nr[NI::R + b * D + a] * da[C_RV + a];
            gv[b] += sv;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                float s = 0.f;
#pragma unroll
                for (int a = 0; a < D; ++a) s += nr[NI::R + b * D + a] * dM[a][c];
                GR[b][c] += s;
            }
        }
    }
    // ---- layer_1.res(rel_feat): columns D..2D-1 (R^T v) and 2D..3D-1 (R^T f) of W_res^T dn_1
    for (int o = t; o < hid; o += 32) {
        const float gg = DN1[nc * hid + o];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            dcv[d] += P.l1_res_w[o * 3 * D + D + d] * gg;
            dcf[d] += P.l1_res_w[o * 3 * D + 2 * D + d] * gg;
        }
    }
#pragma unroll
    for (int b = 0; b < D; ++b) {
        gp[b] = sum32(gp[b]); gv[b] = sum32(gv[b]); dcv[b] = sum32(dcv[b]); dcf[b] = sum32(dcf[b]);
#pragma unroll
        for (int a = 0; a < D; ++a) GR[b][a] = sum32(GR[b][a]);
    }
    // ---- node level: rel_feat = [0 | R^T v | R^T f], out = p + R y
    float y[D], go[D];
#pragma unroll
    for (int a = 0; a < D; ++a) {
        float s = 0.f;
#pragma unroll
        for (int b = 0; b < D; ++b) s += ni[NI::R + b * D + a] * (out[nc * D + b] - ni[NI::P + b]);
        y[a] = s;
        go[a] = g_out[nc * D + a];
    }
#pragma unroll
    for (int b = 0; b < D; ++b) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < D; ++a) {
            GR[b][a] += ni[NI::V + b] * dcv[a] + ni[NI::F + b] * dcf[a] + go[b] * y[a];
            s += ni[NI::R + b * D + a] * dcv[a];
        }
        gv[b] += s;                              // R^T v seen as a function of v with the frame held fixed
        gp[b] += go[b];                          // the residual x + ...
    }
    // ---- frame angles -> velocity
    {
        const float* R = ni + NI::R;
        const float v0 = ni[NI::V], v1 = ni[NI::V + 1];
        const float den = v0 * v0 + v1 * v1;
        float dth = 0.f;
        if constexpr (D == 2) {
            const float c = R[0], s = R[2];
            dth = GR[0][0] * (-s) + GR[0][1] * (-c) + GR[1][0] * c + GR[1][1] * (-s);
        } else {
            const float c = R[4], s = -R[1], cp = R[8], sp = -R[6];
            dth = GR[0][0] * (-cp * s) + GR[0][1] * (-c) + GR[0][2] * (-sp * s) + GR[1][0] * (cp * c) + GR[1][1] * (-s) +
                  GR[1][2] * (sp * c);
            const float dph = GR[0][0] * (-sp * c) + GR[0][2] * (cp * c) + GR[1][0] * (-sp * s) + GR[1][2] * (cp * s) +
                              GR[2][0] * (-cp) + GR[2][2] * (-sp);
            const float v2 = ni[NI::V + 2];
            const float rho = sqrtf(den + v2 * v2);
            const float cz = v2 / (rho + EPS_F);
            if (rho > 0.f && cz > -1.0f && cz < 1.0f) {
                const float dcz = -dph / sqrtf(1.0f - cz * cz);
                gv[2] += dcz / (rho + EPS_F);
                const float k = -dcz * v2 / ((rho + EPS_F) * (rho + EPS_F) * rho);
                gv[0] += k * v0; gv[1] += k * v1; gv[2] += k * v2;
            }
        }
        if (den > 0.f) {
            gv[0] += dth * (-v1 / den);
            gv[1] += dth * (v0 / den);
        }
    }
    if (field_gz != nullptr) {              // (uniform branch: no barrier is skipped by part of a workgroup)
        if (!ok || t >= 2 * D) return;
        const float dzx = field_gz[n * 2 * D + t];
        float resx = 0.f;
#pragma unroll
        for (int b = 0; b < D; ++b) {
            if (t == b) resx = gp[b] + dzx;
            if (t == D + b) resx = gv[b] + dzx;
        }
        if (t < D) grad_x[n * D + t] = resx;
        else grad_v[n * D + t - D] = resx;
        return;
    }
    // ---- the field net's inputs: recompute its backward from dL/df (thread t owns hidden unit t), dz = W0^T dpre1
    long ci = (long)(charges[nc] + 1.0f);
    ci = ci < 0 ? 0 : (ci > 2 ? 2 : ci);
    float zt = 0.f;
    if (t < D) zt = x[nc * D + t];
    else if (t < 2 * D) zt = vel[nc * D + t - D];
    else if (t < FIN) zt = P.field_emb[ci * 16 + t - 2 * D];
    sz[g][t] = zt;
    __syncthreads();
    float p1 = P.field_b0[t];
#pragma unroll
    for (int k = 0; k < FIN; ++k) p1 += P.field_w0[t * FIN + k] * sz[g][k];
    sh1[g][t] = silu(p1);
    __syncthreads();
    float p2 = P.field_b2[t];
#pragma unroll
    for (int k = 0; k < 32; ++k) p2 += P.field_w2[t * 32 + k] * sh1[g][k];
    float d2 = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) d2 += P.field_w4[d * 32 + t] * DF[nc * 16 + d];
    d2 *= dsilu(p2);
    sd2[g][t] = d2;
    __syncthreads();
    float d1 = 0.f;
#pragma unroll
    for (int o = 0; o < 32; ++o) d1 += P.field_w2[o * 32 + t] * sd2[g][o];
    d1 *= dsilu(p1);
    sd1[g][t] = d1;
    __syncthreads();
    if (!ok || t >= 2 * D) return;
    float dz = 0.f;
#pragma unroll 8
    for (int o = 0; o < 32; ++o) dz += P.field_w0[o * FIN + t] * sd1[g][o];
    float res = 0.f;
#pragma unroll
    for (int b = 0; b < D; ++b) {
        if (t == b) res = gp[b] + dz;
        if (t == D + b) res = gv[b] + dz;
    }
    if (t < D) grad_x[n * D + t] = res;
    else grad_v[n * D + t - D] = res;
}

// dL/d(edge_attr_orig) [E][2] in the caller's edge order: the last two feature columns of DA, un-permuted
__global__ void __launch_bounds__(256)
kb_edge_attr_grad(const float* __restrict__ DA, const int32_t* __restrict__ perm, int col0, float* __restrict__ grad_ea,
                  int64_t n_edges) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n_edges) return;
    const int64_t e = perm[k];
    grad_ea[2 * e] = DA[k * FPAD + col0];
    grad_ea[2 * e + 1] = DA[k * FPAD + col0 + 1];
}

}  // namespace
