// K-apply-u, sum-factorised form for 3D Q2 on a uniform box (gfx950, wave64).
//
// On a tensor-product mesh with constant coefficients the elasticity operator of
// PoroElasticDisplacementSolver::assemble_system (:230-242) is EXACTLY a sum of Kronecker products of the 1D banded
// matrices M (mass), K (stiffness) and C (C[m][n] = int phi_m' phi_n) of the three directions:
//   A_aa = (lambda+2G) K_a (x) M (x) M + G sum_{d != a} K_d (x) M (x) M,   A_ab = lambda C_a (x) C_b^T (x) M + G C_a^T (x) C_b (x) M
// (validated against the assembled matrix to 1e-15 in tools/kron_proto.py).  With C = O + D (off-diagonal part + boundary
// diagonal) and C^T = -O + D, one application is 9 z-sweeps + 15 y-sweeps + 9 x-sweeps of <= 5-point 1D stencils:
// ~180 flop-instructions per node instead of ~800 FMAs for the element-matrix gather.
//
// Mapping: a 1024-thread workgroup owns a 60 x 12 node column of the mesh (64 x 16 with halo) and MARCHES along z over a chunk
// of planes; one thread per node, one tile row per wavefront (~100 VGPRs -> 4 waves per SIMD to hide latencies).
//   z: each thread keeps a 5-plane window of its node in registers (sliding, planes processed in vertex/mid pairs);
//   y: the nine z-stage fields of a plane go through LDS (8 B per lane, conflict-free); a wave holds ONE row, so the row
//      parity and with it every band coefficient is wave-uniform (inline constants, no divergence);
//   x: lanes are consecutive nodes; contributions to the nodes at +-1, +-2 travel by wave shuffles (scatter form).
// Rows are dealt to waves so that the four waves sharing a SIMD carry equal work (vertex rows cost more than mid rows,
// halo rows only run the z-stage).  Out-of-domain nodes are loaded as zeros, so only the centre coefficients know about
// the domain boundary.  No atomics; results are bitwise reproducible.  Dirichlet columns are masked as planes enter the window; the
// Dirichlet ROWS hold the unconstrained row sums: PCG never reads them (inert dofs), poro_apply_operator finishes them with
// k_kron_fix_constrained.
#include "common.hpp"
#include <hip/hip_ext.h>
#include <type_traits>

namespace poro {
namespace {

// Two tile shapes share the kernel body: 64 lanes x 16 rows (one row per wave, 60 x 12 valid outputs) and, for the remainder of
// the x-extent, 32 lanes x 32 rows (two rows of equal parity per wave, 28 x 28 valid).  Both hold 1024 nodes per plane.
constexpr int NFLD = 18;         // two buffers of the 9 z-stage fields of a plane
constexpr int kTileNodes = 1024;

// FE_Q(2) 1D element matrices on a cell of length h are small-integer matrices times a scale:
//   M = (h/30) [[4,2,-1],[2,16,2],[-1,2,4]],  K = (1/3h) [[7,-8,1],[-8,16,-8],[1,-8,7]],  C = (1/6) [[-3,-4,1],[4,0,-4],[-1,4,3]]
// (checked against Gauss quadrature of the Lagrange basis, check_q2_element_matrices).  Assembled rows:
//   vertex row: M (-1, 2, cM, 2, -1)  K (1, -8, cK, -8, 1)  O (-1, 4, ., -4, 1)  D = cD;  cM = 4(mL+mR), cK = 7(mL+mR), cD = 3(mL-mR)
//   mid row:    M (2, 16, 2)          K (-8, 16, -8)        O (4, ., -4)         D = 0        (mL / mR: left / right cell exists)
// so every band coefficient is an inline constant and all scales fold into the 18 uniform constants below.
struct KronConsts {
  double xk_l2g, xk_g;                         // (lambda+2G | G) sMy sKx sMz                                   -> XK
  double m_gKyMz, m_gMyKz, m_lKyMz, m_lMyKz;   // G sKy sMx sMz, G sMy sMx sKz, (l+2G) sKy sMx sMz, (l+2G) sMy sMx sKz   -> XM
  double cc_mz[4];                             // c1..c4 * sC sC sMz   (O_y / D_y of mz      -> XO, XD)
  double cc_oz[4];                             // c1..c4 * sMy sC sC   (M_y of oz / wz        -> XO, XD)
  double cc_mx[4];                             // c1..c4 * sC sMx sC   (O_y / D_y of oz / wz  -> XM)
};
struct KronArgs {
  int nn[3]; int n64, nty64, has32, nty32, x0_32, nzc, zunit, zq, zr, nA, nblocks, cols, zmajor;   // z-chunk zc owns zq (+1 if zc < zr) units of zunit planes, the last one also the tail; nA = workgroups with 64-lane tiles
  KronConsts k;
  const uint8_t *nodemask; int constrained, mask_anywhere;
  double *dot_partials;   // optional: per-workgroup partial of x.y over the free rows (x is zero on the Dirichlet columns after masking)
  const PcgScalars *pcg;  // optional: inside PCG the launch is a no-op once the solve has finished (the host enqueues iterations in batches)
  KronCheb cheb;          // CHEB kernels: the Chebyshev recurrence is applied where the product leaves the registers (see kron_tile's emit)
};

__device__ inline int64_t xcd_remap(int64_t bid, int64_t n) {
  const int64_t q = n / 8, r = n % 8, xcd = bid % 8, idx = bid / 8;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// workgroup -> (tile column, z-chunk).  The hardware deals workgroups to the 8 XCDs round-robin; xcd_remap gives every XCD a contiguous range of
// tile numbers.  z-major numbering puts all columns of ONE z-chunk on one XCD (cols = 32, nzc = 8 at 72^3): x- and y-neighbours march in lockstep
// through the same planes and find each other's halo rows / lines in that XCD's L2.  (Column-major numbering shares only among 4 y-neighbours.)
__device__ inline void tile_of(const KronArgs &a, int &col, int &zc) {
  const int tile = (int)xcd_remap(blockIdx.x, a.nblocks);
  if (a.zmajor) { zc = tile / a.cols; col = tile % a.cols; }
  else if (tile < a.nA) { zc = tile % a.nzc; col = tile / a.nzc; }
  else { const int t = tile - a.nA; zc = t % a.nzc; col = a.n64 * a.nty64 + t / a.nzc; }
}

// whole-wave lane shifts by DPP (VALU, no LDS round trip): lane l receives the value of lane l-1 (up) / l+1 (down); the lanes at the
// wave edge receive 0 (bound_ctrl) - they are tile-halo lanes whose results are never stored
__device__ inline double wave_up1(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ inline double wave_dn1(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// rows of the tile handled by wave w (both of one parity); waves w, w+4, w+8, w+12 share a SIMD, so the tables give every SIMD
// a similar mix of vertex rows (5-point y-band), mid rows (3-point) and halo rows (z-stage only)
__device__ const signed char kRows64[16] = {2, 4, 6, 8, 3, 5, 7, 9, 10, 11, 12, 13, 1, 0, 15, 14};
#define PORO_ROWS32A {2, 6, 10, 14, 3, 7, 11, 15, 18, 22, 26, 27, 19, 23, 0, 1}
#define PORO_ROWS32B {4, 8, 12, 16, 5, 9, 13, 17, 20, 24, 28, 29, 21, 25, 30, 31}
__device__ const signed char kRows32a[16] = PORO_ROWS32A;
__device__ const signed char kRows32b[16] = PORO_ROWS32B;
// the plane loop is specialised on the row parity of a wave and holds workgroup barriers inside both instantiations: the two rows of a wave must have the same parity
constexpr bool rows32_same_parity() { constexpr signed char a[16] = PORO_ROWS32A, b[16] = PORO_ROWS32B; for (int i = 0; i < 16; ++i) if ((a[i] ^ b[i]) & 1) return false; return true; }
static_assert(rows32_same_parity(), "kRows32a / kRows32b: the two rows of a wave differ in parity");

// hardware-bounds-checked buffer access (raw buffer, stride 0): a lane whose byte offset lies outside the buffer loads 0 / stores nothing, so
// the tile halo, the domain edge and the Dirichlet-mask conditions become OFFSETS instead of branches and the plane body stays one basic block
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;            // beyond any buffer this kernel accepts (kron_apply checks the size)
__device__ inline void bload3(__amdgpu_buffer_rsrc_t r, unsigned off, double (&v)[3]) {
  const v4u t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); const v2u u = __builtin_amdgcn_raw_buffer_load_b64(r, off + 16u, 0, 0);
  v[0] = __hiloint2double(t.y, t.x); v[1] = __hiloint2double(t.w, t.z); v[2] = __hiloint2double(u.y, u.x);
}
__device__ inline double bload1(__amdgpu_buffer_rsrc_t r, unsigned off) { const v2u u = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); return __hiloint2double(u.y, u.x); }
__device__ inline void bstore1(__amdgpu_buffer_rsrc_t r, unsigned off, double v) { v2u u; u.x = __double2loint(v); u.y = __double2hiint(v); __builtin_amdgcn_raw_buffer_store_b64(u, r, off, 0, 0); }

template <int TXN, int CHEB>   // CHEB: 0 plain operator, 1 fused Chebyshev update, 2 the same + raw partial product on the first / last plane (slab partitions)
__device__ __forceinline__ void kron_tile(const KronArgs &a, const double *__restrict__ x, double *__restrict__ y, double *L, const int X0, const int tyi, const int zc) {
  constexpr int RPW = 64 / TXN, TYR = 16 * RPW, VY = TYR - 4;   // rows per wave, rows per tile, valid rows
  static_assert(TXN * TYR == kTileNodes, "tile");
  const int tid = threadIdx.x;
  const int NX = a.nn[0], NY = a.nn[1], NZ = a.nn[2];
  const int Y0 = VY * tyi - 2;
  const int k0 = a.zunit * (zc * a.zq + min(zc, a.zr)), k1 = zc == a.nzc - 1 ? NZ : a.zunit * ((zc + 1) * a.zq + min(zc + 1, a.zr));

  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lx = lane % TXN;
  const int ra = RPW == 1 ? kRows64[w] : kRows32a[w], rb = RPW == 1 ? ra : kRows32b[w];
  const int r = (RPW == 2 && lane >= TXN) ? rb : ra;
  const bool odd_row = (__builtin_amdgcn_readfirstlane(ra) & 1) != 0, halo_wave = ra < 2 || ra > TYR - 3;   // (scalar: the parity branch below is wave-uniform by construction, and the barriers inside it are reached by every wave in the same order)
  const int j = Y0 + r, i = X0 + lx;
  const bool even_i = (lx & 1) == 0;
  const bool vj = j >= 0 && j < NY, vn = vj && i >= 0 && i < NX;
  const bool out = vn && lx >= 2 && lx <= TXN - 3 && !halo_wave;
  const bool bnd_xy = i == 0 || i == NX - 1 || j == 0 || j == NY - 1;

  // the descriptors cover the planes THIS z-chunk touches (pbase .. pend - 1), so 32-bit offsets suffice for vectors of any length (kron_apply keeps a chunk's
  // span below 2^31 bytes); a prefetch beyond pend is out of range and returns zeros like every other predicate
  const unsigned nxy = (unsigned)NX * (unsigned)NY;
  const int pbase = max(k0 - 2, 0), pend = min(k1 + 2, NZ);
  const unsigned span_nodes = (unsigned)(pend - pbase) * nxy;
  const size_t base_node = (size_t)pbase * nxy;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(x + 3 * base_node), 0, span_nodes * 24u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)((CHEB ? a.cheb.znew : y) + 3 * base_node), 0, span_nodes * 24u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)(a.nodemask + base_node), 0, a.constrained ? span_nodes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)(a.cheb.g + (CHEB ? 3 * base_node : 0)), 0, CHEB ? span_nodes * 24u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.cheb.cls + (CHEB ? base_node : 0)), 0, CHEB ? span_nodes : 0u, 0x00020000);
  const unsigned nxy_off = vn ? (unsigned)(j * NX + i) : kOOB;          // node offset inside a plane (invalid lanes: out of every buffer)
  const unsigned out_off = out ? (unsigned)(j * NX + i) : kOOB;

  // centre coefficients (the only place the domain boundary enters); pe switches the +-2 messages off for mid nodes
  const double mLx = i > 0 ? 1.0 : 0.0, mRx = i < NX - 1 ? 1.0 : 0.0;
  const double pe = even_i ? 1.0 : 0.0;
  const double cMx = even_i ? 4.0 * (mLx + mRx) : 16.0, cKx = even_i ? 7.0 * (mLx + mRx) : 16.0, cDx = even_i ? 3.0 * (mLx - mRx) : 0.0;
  // vertex rows (mid rows: 16, 16, 0 as literals); selected from literals so that a wave holding ONE row (64-lane tile) keeps them in scalar registers
  const bool inner_y = j > 0 && j < NY - 1;
  const double cMyv = inner_y ? 8.0 : 4.0, cKyv = inner_y ? 14.0 : 7.0, cDyv = inner_y ? 0.0 : (j > 0 ? 3.0 : -3.0);
  const KronConsts &K = a.k;

  // loads are asynchronous: the node's constraint bits (bit c = dof (node,c) is a Dirichlet dof) travel with the values and are
  // applied only when the plane enters the window, so no wait on the in-flight prefetch is ever forced early.  Planes outside the box,
  // nodes outside the domain and nodes whose mask cannot be set are out-of-range offsets: they load zeros without a branch
  const bool mask_all = a.mask_anywhere != 0;
  auto load_plane = [&](int p, double (&v)[3], unsigned &m) {
    const bool pin = p >= 0 && p < NZ;                                   // wave-uniform
    const unsigned node = (unsigned)(p - pbase) * nxy + nxy_off;       // relative to the chunk's first plane
    const unsigned off = (pin && vn) ? node * 24u : kOOB;
    bload3(rx, off, v);
    const bool mk = pin && vn && (mask_all || p == 0 || p == NZ - 1 || bnd_xy);
    m = __builtin_amdgcn_raw_buffer_load_b8(rm, mk ? node : kOOB, 0, 0);
  };
  auto apply_mask = [&](double (&v)[3], unsigned m) {
    if (__builtin_amdgcn_ballot_w64(m != 0) == 0) return;      // no Dirichlet dof in this wave's row of the plane (every interior tile): skip the selects
#pragma unroll
    for (int c = 0; c < 3; ++c) if (m & (1u << c)) v[c] = 0.0;
  };

  double W0[3], W1[3], W2[3], W3[3], W4[3];
  unsigned m0, m1;
  load_plane(k0 - 2, W0, m0); apply_mask(W0, m0); load_plane(k0 - 1, W1, m0); apply_mask(W1, m0); load_plane(k0, W2, m0); apply_mask(W2, m0);
  load_plane(k0 + 1, W3, m0); apply_mask(W3, m0); load_plane(k0 + 2, W4, m0); apply_mask(W4, m0);

  // one plane: z-stage in registers -> LDS (double buffered: ONE barrier per plane) -> y-stage -> x-stage -> store
  int buf = 0;
  double dot_acc = 0.0;
  auto plane = [&](const int kk, const bool oddz, auto &&after_zstage) {
    double *Lw = L + buf * (9 * TYR * TXN);
    {
      double f[9];       // mz_x mz_y mz_z kz_x kz_y kz_z oz_x oz_y oz_z   (unscaled: integer band coefficients)
      if (!oddz) {
        const double mL = kk > 0 ? 1.0 : 0.0, mR = kk < NZ - 1 ? 1.0 : 0.0;
        const double cM = 4.0 * (mL + mR), cK = 7.0 * (mL + mR);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double t1 = W1[c] + W3[c], t2 = W0[c] + W4[c];
          f[c] = fma(cM, W2[c], fma(2.0, t1, -t2));
          f[3 + c] = fma(cK, W2[c], fma(-8.0, t1, t2));
          f[6 + c] = fma(4.0, W1[c] - W3[c], W4[c] - W0[c]);
        }
      } else {   // mid plane: couples to kk-1, kk, kk+1 = W2, W3, W4
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double t1 = W2[c] + W4[c];
          f[c] = fma(16.0, W3[c], 2.0 * t1);
          f[3 + c] = fma(16.0, W3[c], -8.0 * t1);
          f[6 + c] = 4.0 * (W2[c] - W4[c]);
        }
      }
      // safe without a barrier: every wave passed the previous barrier only after it finished reading this buffer two planes ago
#pragma unroll
      for (int q = 0; q < 9; ++q) Lw[(q * TYR + r) * TXN + lx] = f[q];
    }
    after_zstage();                                    // (the even plane issues the prefetch of the next pair here: W0, W1 are dead now)
#ifndef PORO_DIAG_NO_BARRIER
    __syncthreads();
#endif
    const double *Lb = Lw; buf ^= 1;

    const bool has_w = !oddz && (kk == 0 || kk == NZ - 1);   // workgroup-uniform: first / last plane of the box
    const double (&xc)[3] = oddz ? W3 : W2;            // this plane's (masked) input values, for the fused x.y
    // everything behind the barrier is specialised on the row parity of the wave (one wave-uniform branch per plane instead of one per field):
    // mid rows have 3-point y-bands and no boundary diagonal, vertex rows 5-point bands
    auto rest = [&](auto ODD) {
      constexpr bool odd = decltype(ODD)::value;
      // y-stage helpers: neighbours and the node's own value of field q come from the LDS buffer `Lb`
      double s1, s2 = 0.0, d1, d2 = 0.0, own;   // v(-1)+v(+1), v(-2)+v(+2), v(-1)-v(+1), v(+2)-v(-2), v(0)
      auto nb = [&](int q) {
        const double *col = Lb + (q * TYR + r) * TXN + lx;
        const double nm1 = col[-TXN], np1 = col[TXN];
        own = col[0];
        s1 = nm1 + np1; d1 = nm1 - np1;
        if constexpr (!odd) { const double nm2 = col[-2 * TXN], np2 = col[2 * TXN]; s2 = nm2 + np2; d2 = np2 - nm2; }
      };
      auto sweepM = [&]() { if constexpr (odd) return fma(2.0, s1, 16.0 * own); else return fma(2.0, s1, cMyv * own) - s2; };
      auto sweepK = [&]() { if constexpr (odd) return fma(-8.0, s1, 16.0 * own); else return fma(-8.0, s1, cKyv * own) + s2; };
      auto sweepO = [&]() { if constexpr (odd) return 4.0 * d1; else return fma(4.0, d1, d2); };
      // c * D_y v + rest (D_y = boundary diagonal of C_y: vertex rows on the first / last row of the box only, never on mid rows)
      auto fmaD = [&](double c, double rest_) { if constexpr (odd) return rest_; else return fma(c, cDyv * own, rest_); };

      // x-stage of one component, scatter form: the node's own term + messages to the nodes at +-1 and (vertex nodes only) +-2
      auto xstage = [&](const double FK, const double FM, const double FO, const double FD) {
        const double t1 = fma(-8.0, FK, 2.0 * FM);         // K / M part of the +-1 coupling (the same for vertex and mid sources)
        const double t2 = pe * (FK - FM);                  // +-2 coupling exists only between vertex nodes
        const double pO = pe * FO;
        double sacc = fma(cKx, FK, fma(cMx, FM, cDx * FD));
#ifndef PORO_DIAG_NO_SHUFFLE
        // from i-1 (its +1 message) and i+1 (its -1 message); the messages from i-2 / i+2 ride along: shifted once on their own, then with the +-1 message
        sacc += wave_up1(fma(4.0, FO, t1) + wave_up1(t2 - pO)) + wave_dn1(fma(-4.0, FO, t1) + wave_dn1(t2 + pO));
#else
        sacc += fma(4.0, FO, t1) + fma(-4.0, FO, t1) + (t2 - pO) + (t2 + pO);
#endif
        return sacc;
      };
      const unsigned d0b = ((unsigned)(kk - pbase) * nxy + out_off) * 24u;       // byte offset of the node's dofs in the chunk; lanes without an output: out of range
      const unsigned so = out ? d0b : kOOB;
      // CHEB: x is the iterate z_j of the polynomial preconditioner in root form; instead of A z_j the kernel stores z_{j+1} = z_j + omega_j D^-1 (g - A z_j)
      // (omega_j = reciprocal of a root of the shifted Chebyshev polynomial; z_{j+1} goes to the other buffer of a ping-pong pair because neighbouring
      // tiles still read z_j) and accumulates g . z_{j+1}.  One extra read stream (g) instead of the two of the three-term recurrence.  Dirichlet dofs have
      // D^-1 = 0 and z = 0.  Lanes without an output load g = 0 and store nothing.
      const double *ctab = nullptr;
      if constexpr (CHEB) ctab = a.cheb.tab + 3u * __builtin_amdgcn_raw_buffer_load_b8(rc, out ? (unsigned)(kk - pbase) * nxy + out_off : kOOB, 0, 0);
      auto emit = [&](int c, double v) {
        if constexpr (CHEB) {
          const double gi = bload1(rg, so + 8u * c);
          const double zn = fma(a.cheb.omega * ctab[c], gi - v, xc[c]);
          bstore1(ry, so + 8u * c, zn); dot_acc = fma(gi, zn, dot_acc);
        } else { bstore1(ry, so + 8u * c, v); dot_acc = fma(xc[c], v, dot_acc); }   // Dirichlet rows: see the header; lanes without an output are dropped from the sum at the end
      };

      if (!has_w) {
        if (halo_wave) return;
        // field order chosen so that the inputs of component x, then y, then z complete early and their registers die.  The LDS reads of the NEXT
        // field are issued before the arithmetic of the current one (two register sets, software-pipelined by hand: the compiler keeps the order)
        struct Nb { double nm2, nm1, own, np1, np2; };
        auto ld = [&](int q, Nb &n) {
          const double *col = Lb + (q * TYR + r) * TXN + lx;
          n.nm1 = col[-TXN]; n.np1 = col[TXN]; n.own = col[0];
          if constexpr (!odd) { n.nm2 = col[-2 * TXN]; n.np2 = col[2 * TXN]; }
        };
        auto use = [&](const Nb &n) { own = n.own; s1 = n.nm1 + n.np1; d1 = n.nm1 - n.np1; if constexpr (!odd) { s2 = n.nm2 + n.np2; d2 = n.np2 - n.nm2; } };
        Nb A, B;
        double XK0, XM0, XO0, XD0, XK1, XM1, XO1, XD1;
        ld(0, A); ld(3, B);
        { use(A); const double My = sweepM(), Ky = sweepK(), Oy = sweepO();                  // mz_x
          XK0 = K.xk_l2g * My; XM0 = K.m_gKyMz * Ky; XO1 = fmaD(K.cc_mz[2], K.cc_mz[0] * Oy); XD1 = fmaD(K.cc_mz[3], K.cc_mz[1] * Oy); }
        ld(1, A);
        { use(B); XM0 = fma(K.m_gMyKz, sweepM(), XM0); }                                     // kz_x
        ld(8, B);
        { use(A); const double My = sweepM(), Ky = sweepK(), Oy = sweepO();                  // mz_y
          XK1 = K.xk_g * My; XM1 = K.m_lKyMz * Ky; XO0 = fmaD(K.cc_mz[1], K.cc_mz[0] * Oy); XD0 = fmaD(K.cc_mz[3], K.cc_mz[2] * Oy); }
        ld(4, A);
        { use(B); const double My = sweepM(), Oy = sweepO();                                 // oz_z
          XO0 = fma(K.cc_oz[0], My, XO0); XD0 = fma(K.cc_oz[2], My, XD0); XM1 = fma(K.cc_mx[0], Oy, fmaD(K.cc_mx[2], XM1)); }
        ld(2, B);
        emit(0, xstage(XK0, XM0, XO0, XD0));
        { use(A); XM1 = fma(K.m_gMyKz, sweepM(), XM1); }                                     // kz_y
        ld(5, A);
        emit(1, xstage(XK1, XM1, XO1, XD1));
        double XK2, XM2, XO2, XD2;
        { use(B); const double My = sweepM(), Ky = sweepK(); XK2 = K.xk_g * My; XM2 = K.m_gKyMz * Ky; }   // mz_z
        ld(6, B);
        { use(A); XM2 = fma(K.m_lMyKz, sweepM(), XM2); }                                     // kz_z
        ld(7, A);
        { use(B); const double My = sweepM(); XO2 = K.cc_oz[0] * My; XD2 = K.cc_oz[1] * My; } // oz_x
        { use(A); const double Oy = sweepO(); XM2 = fma(K.cc_mx[0], Oy, fmaD(K.cc_mx[1], XM2)); }   // oz_y
        emit(2, xstage(XK2, XM2, XO2, XD2));
        return;
      }

      // first / last plane of the box (two planes per box): additionally the boundary diagonal D_z of C_z, wz = cD u, which goes
      // through slots 0..2 of the same buffer in a second round
      double XK[3], XM[3], XO[3], XD[3];
      const double cD = 3.0 * ((kk > 0 ? 1.0 : 0.0) - (kk < NZ - 1 ? 1.0 : 0.0));
      if (!halo_wave) {
        { nb(0); const double My = sweepM(), Ky = sweepK(), Oy = sweepO();                   // mz_x
          XK[0] = K.xk_l2g * My; XM[0] = K.m_gKyMz * Ky; XO[1] = fmaD(K.cc_mz[2], K.cc_mz[0] * Oy); XD[1] = fmaD(K.cc_mz[3], K.cc_mz[1] * Oy); }
        { nb(1); const double My = sweepM(), Ky = sweepK(), Oy = sweepO();                   // mz_y
          XK[1] = K.xk_g * My; XM[1] = K.m_lKyMz * Ky; XO[0] = fmaD(K.cc_mz[1], K.cc_mz[0] * Oy); XD[0] = fmaD(K.cc_mz[3], K.cc_mz[2] * Oy); }
        { nb(2); const double My = sweepM(), Ky = sweepK(); XK[2] = K.xk_g * My; XM[2] = K.m_gKyMz * Ky; }   // mz_z
        { nb(3); XM[0] = fma(K.m_gMyKz, sweepM(), XM[0]); }                                    // kz_x
        { nb(4); XM[1] = fma(K.m_gMyKz, sweepM(), XM[1]); }                                    // kz_y
        { nb(5); XM[2] = fma(K.m_lMyKz, sweepM(), XM[2]); }                                    // kz_z
        { nb(6); const double My = sweepM(); XO[2] = K.cc_oz[0] * My; XD[2] = K.cc_oz[1] * My; }   // oz_x
        { nb(7); const double Oy = sweepO(); XM[2] = fma(K.cc_mx[0], Oy, fmaD(K.cc_mx[1], XM[2])); }   // oz_y
        { nb(8); const double My = sweepM(), Oy = sweepO();                                    // oz_z
          XO[0] = fma(K.cc_oz[0], My, XO[0]); XD[0] = fma(K.cc_oz[2], My, XD[0]); XM[1] = fma(K.cc_mx[0], Oy, fmaD(K.cc_mx[2], XM[1])); }
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 3; ++c) Lw[(c * TYR + r) * TXN + lx] = cD * xc[c];
      __syncthreads();
      if (!halo_wave) {
        { nb(0); const double My = sweepM(); XO[2] = fma(K.cc_oz[2], My, XO[2]); XD[2] = fma(K.cc_oz[3], My, XD[2]); }            // wz_x
        { nb(1); const double Oy = sweepO(); XM[2] = fma(K.cc_mx[2], Oy, fmaD(K.cc_mx[3], XM[2])); }                              // wz_y
        { nb(2); const double My = sweepM(), Oy = sweepO();                                                                       // wz_z
          XO[0] = fma(K.cc_oz[1], My, XO[0]); XD[0] = fma(K.cc_oz[3], My, XD[0]); XM[1] = fma(K.cc_mx[1], Oy, fmaD(K.cc_mx[3], XM[1])); }
      }
      __syncthreads();   // slots 0..2 must not be overwritten by a fast wave's plane kk+2 before everybody has read wz (same buffer)
      if (halo_wave) return;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double v = xstage(XK[c], XM[c], XO[c], XD[c]);
        if constexpr (CHEB == 2) {   // slab partitions: the raw partial product on the planes shared with the neighbours goes to a side buffer as well (ctx.hip, fuse_multi)
          double *sd = kk == 0 ? a.cheb.side_lo : a.cheb.side_hi;
          if (sd && out) sd[3 * (j * NX + i) + c] = v;
        }
        emit(c, v);
      }
    };
    if (odd_row) rest(std::true_type{}); else rest(std::false_type{});
  };

  for (int k = k0; k < k1; k += 2) {
    plane(k, false, [&] { load_plane(k + 3, W0, m0); load_plane(k + 4, W1, m1); });   // planes k-2, k-1 are no longer needed
    if (k + 1 < k1) plane(k + 1, true, [] {});
    // rotate the window by two planes; the prefetched planes get their Dirichlet columns zeroed as they enter
#pragma unroll
    for (int c = 0; c < 3; ++c) { const double t0 = W0[c], t1 = W1[c]; W0[c] = W2[c]; W1[c] = W3[c]; W2[c] = W4[c]; W3[c] = t0; W4[c] = t1; }
    apply_mask(W3, m0); apply_mask(W4, m1);
  }
  if (a.dot_partials) {   // deterministic workgroup reduction of x.y (fixed shuffle tree, waves summed in index order)
    if (!out) dot_acc = 0.0;                                   // halo lanes / rows accumulated finite garbage
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_xor(dot_acc, off, 64);
    __syncthreads();
    if (lane == 0) L[w] = dot_acc;
    __syncthreads();
    if (tid == 0) { double t = 0; for (int q = 0; q < 16; ++q) t += L[q]; a.dot_partials[blockIdx.x] = t; }
  }
}

__global__ void __launch_bounds__(1024)
k_kron3_q2(KronArgs a, const double *__restrict__ x, double *__restrict__ y) {
  extern __shared__ double L[];                            // [2 buffers][9 fields][1024 nodes of the tile plane]
  if (a.pcg && (a.pcg->done || a.pcg->finishing)) return;  // uniform over the grid: written by the previous launches only
  int col, zc; tile_of(a, col, zc);                        // workgroup-uniform: either tile shape, never both
  if (col < a.n64 * a.nty64) kron_tile<64, 0>(a, x, y, L, 60 * (col / a.nty64) - 2, col % a.nty64, zc);
  else kron_tile<32, 0>(a, x, y, L, a.x0_32, col - a.n64 * a.nty64, zc);
}
// the same sweep with the Chebyshev (root form) update fused into the stores (polynomial preconditioner of the displacement CG: no vector kernels and no
// reductions between the operator applications of one preconditioner call)
__global__ void __launch_bounds__(1024)
k_kron3_q2_cheb(KronArgs a, const double *__restrict__ x, double *__restrict__ y) {
  extern __shared__ double L[];
  if (a.pcg && (a.pcg->done || a.pcg->finishing)) return;
  int col, zc; tile_of(a, col, zc);                        // workgroup-uniform: either tile shape, never both
  if (col < a.n64 * a.nty64) kron_tile<64, 1>(a, x, y, L, 60 * (col / a.nty64) - 2, col % a.nty64, zc);
  else kron_tile<32, 1>(a, x, y, L, a.x0_32, col - a.n64 * a.nty64, zc);
}
// slab partitions: additionally the raw partial product on the two shared node planes (KronCheb::side_lo / side_hi)
__global__ void __launch_bounds__(1024)
k_kron3_q2_cheb_side(KronArgs a, const double *__restrict__ x, double *__restrict__ y) {
  extern __shared__ double L[];
  if (a.pcg && (a.pcg->done || a.pcg->finishing)) return;
  int col, zc; tile_of(a, col, zc);                        // workgroup-uniform: either tile shape, never both
  if (col < a.n64 * a.nty64) kron_tile<64, 2>(a, x, y, L, 60 * (col / a.nty64) - 2, col % a.nty64, zc);
  else kron_tile<32, 2>(a, x, y, L, a.x0_32, col - a.n64 * a.nty64, zc);
}


// ---- 3D Q1: the same construction with 3-point bands ---------------------------------------------------------------------------
// FE_Q(1) 1D element matrices: M = (h/6) [[2,1],[1,2]], K = (1/h) [[1,-1],[-1,1]], C = (1/2) [[-1,-1],[1,1]]; assembled rows
//   M (1, cM, 1), cM = 2(mL+mR);  K (-1, cK, -1), cK = mL+mR;  O (1, ., -1);  D = mL - mR.
// Every node is a vertex node: 3-plane register window, one plane per barrier, halo of one node / row / plane.  Tiles: 64 lanes x 16
// rows (62 x 14 valid) and 32 lanes x 32 rows (30 x 30 valid).
template <int TXN, int CHEB>
__device__ __forceinline__ void kron_tile_q1(const KronArgs &a, const double *__restrict__ x, double *__restrict__ y, double *L, const int X0, const int tyi, const int zc) {
  constexpr int RPW = 64 / TXN, TYR = 16 * RPW, VY = TYR - 2;
  const int tid = threadIdx.x;
  const int NX = a.nn[0], NY = a.nn[1], NZ = a.nn[2];
  const int Y0 = VY * tyi - 1;
  const int k0 = a.zunit * (zc * a.zq + min(zc, a.zr)), k1 = zc == a.nzc - 1 ? NZ : a.zunit * ((zc + 1) * a.zq + min(zc + 1, a.zr));
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lx = lane % TXN;
  const int r = RPW == 1 ? w : w + 16 * (lane / TXN);
  const bool halo_wave = RPW == 1 && (w == 0 || w == 15);        // (two rows per wave: halo rows share waves with valid rows)
  const int j = Y0 + r, i = X0 + lx;
  const bool vj = j >= 0 && j < NY, vn = vj && i >= 0 && i < NX;
  const bool out = vn && lx >= 1 && lx <= TXN - 2 && r >= 1 && r <= TYR - 2;
  const bool bnd_xy = i == 0 || i == NX - 1 || j == 0 || j == NY - 1;
  const double mLx = i > 0 ? 1.0 : 0.0, mRx = i < NX - 1 ? 1.0 : 0.0, mLy = j > 0 ? 1.0 : 0.0, mRy = j < NY - 1 ? 1.0 : 0.0;
  const double cMx = 2.0 * (mLx + mRx), cKx = mLx + mRx, cDx = mLx - mRx, cMy = 2.0 * (mLy + mRy), cKy = mLy + mRy, cDy = mLy - mRy;
  const KronConsts &K = a.k;

  // bounds-checked buffer access as in kron_tile: halo / out-of-domain / unmasked conditions are out-of-range offsets, not branches
  const unsigned nxy = (unsigned)NX * (unsigned)NY;
  const int pbase = max(k0 - 1, 0), pend = min(k1 + 1, NZ);          // descriptors relative to the chunk's planes (see kron_tile)
  const unsigned span_nodes = (unsigned)(pend - pbase) * nxy;
  const size_t base_node = (size_t)pbase * nxy;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void *)(x + 3 * base_node), 0, span_nodes * 24u, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void *)((CHEB ? a.cheb.znew : y) + 3 * base_node), 0, span_nodes * 24u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc((void *)(a.nodemask + base_node), 0, a.constrained ? span_nodes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void *)(a.cheb.g + (CHEB ? 3 * base_node : 0)), 0, CHEB ? span_nodes * 24u : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)(a.cheb.cls + (CHEB ? base_node : 0)), 0, CHEB ? span_nodes : 0u, 0x00020000);
  const unsigned nxy_off = vn ? (unsigned)(j * NX + i) : kOOB, out_off = out ? (unsigned)(j * NX + i) : kOOB;
  const bool mask_all = a.mask_anywhere != 0;
  auto load_plane = [&](int p, double (&v)[3], unsigned &m) {
    const bool pin = p >= 0 && p < NZ;
    const unsigned node = (unsigned)(p - pbase) * nxy + nxy_off;
    bload3(rx, (pin && vn) ? node * 24u : kOOB, v);
    const bool mk = pin && vn && (mask_all || p == 0 || p == NZ - 1 || bnd_xy);
    m = __builtin_amdgcn_raw_buffer_load_b8(rm, mk ? node : kOOB, 0, 0);
  };
  auto apply_mask = [&](double (&v)[3], unsigned m) {
    if (__builtin_amdgcn_ballot_w64(m != 0) == 0) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) if (m & (1u << c)) v[c] = 0.0;
  };
  double W0[3], W1[3], W2[3];   // planes k-1, k, k+1
  unsigned m0;
  load_plane(k0 - 1, W0, m0); apply_mask(W0, m0); load_plane(k0, W1, m0); apply_mask(W1, m0); load_plane(k0 + 1, W2, m0); apply_mask(W2, m0);

  const double *Lb = L;
  double s1, d1, own;
  auto nb = [&](int q) {
    const double *col = Lb + (q * TYR + r) * TXN + lx;
    const double nm1 = col[-TXN], np1 = col[TXN];
    own = col[0]; s1 = nm1 + np1; d1 = nm1 - np1;
  };
  auto sweepM = [&]() { return fma(cMy, own, s1); };
  auto sweepK = [&]() { return fma(cKy, own, -s1); };
  auto sweepO = [&]() { return d1; };
  int buf = 0;
  double dot_acc = 0.0;

  for (int kk = k0; kk < k1; ++kk) {
    double *Lw = L + buf * (9 * TYR * TXN);
    const double mL = kk > 0 ? 1.0 : 0.0, mR = kk < NZ - 1 ? 1.0 : 0.0;
    {
      const double cM = 2.0 * (mL + mR), cK = mL + mR;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double t1 = W0[c] + W2[c];
        Lw[(c * TYR + r) * TXN + lx] = fma(cM, W1[c], t1);              // mz
        Lw[((3 + c) * TYR + r) * TXN + lx] = fma(cK, W1[c], -t1);       // kz
        Lw[((6 + c) * TYR + r) * TXN + lx] = W0[c] - W2[c];             // oz
      }
    }
    const bool has_w = kk == 0 || kk == NZ - 1;
    const double xc[3] = {W1[0], W1[1], W1[2]};   // this plane's (masked) input values
    // plane k-1 is no longer needed: prefetch plane k+2 into its registers, rotate when the plane is done
    unsigned mp; load_plane(kk + 2, W0, mp);
    __syncthreads();
    Lb = Lw; buf ^= 1;

    auto xstage = [&](const double FK, const double FM, const double FO, const double FD) {
      const double t1 = FM - FK;
      return fma(cKx, FK, fma(cMx, FM, cDx * FD)) + wave_up1(t1 + FO) + wave_dn1(t1 - FO);   // from i-1 and i+1
    };
    const unsigned so = out ? ((unsigned)(kk - pbase) * nxy + out_off) * 24u : kOOB;
    const double *ctab = nullptr;
    if constexpr (CHEB) ctab = a.cheb.tab + 3u * __builtin_amdgcn_raw_buffer_load_b8(rc, out ? (unsigned)(kk - pbase) * nxy + out_off : kOOB, 0, 0);
    auto emit = [&](int c, double v) {
      if constexpr (CHEB) {      // see kron_tile
        const double gi = bload1(rg, so + 8u * c);
        const double zn = fma(a.cheb.omega * ctab[c], gi - v, xc[c]);
        bstore1(ry, so + 8u * c, zn); dot_acc = fma(gi, zn, dot_acc);
      } else { bstore1(ry, so + 8u * c, v); dot_acc = fma(xc[c], v, dot_acc); }
    };

    if (!has_w) {
      if (!halo_wave) {
        double XK0, XM0, XO0, XD0, XK1, XM1, XO1, XD1;
        { nb(0); const double My = sweepM(), Ky = sweepK(), Oy = sweepO(), Dy = cDy * own;
          XK0 = K.xk_l2g * My; XM0 = K.m_gKyMz * Ky; XO1 = fma(K.cc_mz[2], Dy, K.cc_mz[0] * Oy); XD1 = fma(K.cc_mz[3], Dy, K.cc_mz[1] * Oy); }
        { nb(3); XM0 = fma(K.m_gMyKz, sweepM(), XM0); }
        { nb(1); const double My = sweepM(), Ky = sweepK(), Oy = sweepO(), Dy = cDy * own;
          XK1 = K.xk_g * My; XM1 = K.m_lKyMz * Ky; XO0 = fma(K.cc_mz[1], Dy, K.cc_mz[0] * Oy); XD0 = fma(K.cc_mz[3], Dy, K.cc_mz[2] * Oy); }
        { nb(8); const double My = sweepM(), Oy = sweepO(), Dy = cDy * own;
          XO0 = fma(K.cc_oz[0], My, XO0); XD0 = fma(K.cc_oz[2], My, XD0); XM1 = fma(K.cc_mx[0], Oy, fma(K.cc_mx[2], Dy, XM1)); }
        emit(0, xstage(XK0, XM0, XO0, XD0));
        { nb(4); XM1 = fma(K.m_gMyKz, sweepM(), XM1); }
        emit(1, xstage(XK1, XM1, XO1, XD1));
        double XK2, XM2, XO2, XD2;
        { nb(2); const double My = sweepM(), Ky = sweepK(); XK2 = K.xk_g * My; XM2 = K.m_gKyMz * Ky; }
        { nb(5); XM2 = fma(K.m_lMyKz, sweepM(), XM2); }
        { nb(6); const double My = sweepM(); XO2 = K.cc_oz[0] * My; XD2 = K.cc_oz[1] * My; }
        { nb(7); const double Oy = sweepO(), Dy = cDy * own; XM2 = fma(K.cc_mx[0], Oy, fma(K.cc_mx[1], Dy, XM2)); }
        emit(2, xstage(XK2, XM2, XO2, XD2));
      }
    } else {   // first / last plane of the box: additionally the boundary diagonal D_z = mL - mR of C_z, through slots 0..2 in a second round
      double XK[3], XM[3], XO[3], XD[3];
      const double cD = mL - mR;
      if (!halo_wave) {
        { nb(0); const double My = sweepM(), Ky = sweepK(), Oy = sweepO(), Dy = cDy * own;
          XK[0] = K.xk_l2g * My; XM[0] = K.m_gKyMz * Ky; XO[1] = fma(K.cc_mz[2], Dy, K.cc_mz[0] * Oy); XD[1] = fma(K.cc_mz[3], Dy, K.cc_mz[1] * Oy); }
        { nb(1); const double My = sweepM(), Ky = sweepK(), Oy = sweepO(), Dy = cDy * own;
          XK[1] = K.xk_g * My; XM[1] = K.m_lKyMz * Ky; XO[0] = fma(K.cc_mz[1], Dy, K.cc_mz[0] * Oy); XD[0] = fma(K.cc_mz[3], Dy, K.cc_mz[2] * Oy); }
        { nb(2); const double My = sweepM(), Ky = sweepK(); XK[2] = K.xk_g * My; XM[2] = K.m_gKyMz * Ky; }
        { nb(3); XM[0] = fma(K.m_gMyKz, sweepM(), XM[0]); }
        { nb(4); XM[1] = fma(K.m_gMyKz, sweepM(), XM[1]); }
        { nb(5); XM[2] = fma(K.m_lMyKz, sweepM(), XM[2]); }
        { nb(6); const double My = sweepM(); XO[2] = K.cc_oz[0] * My; XD[2] = K.cc_oz[1] * My; }
        { nb(7); const double Oy = sweepO(), Dy = cDy * own; XM[2] = fma(K.cc_mx[0], Oy, fma(K.cc_mx[1], Dy, XM[2])); }
        { nb(8); const double My = sweepM(), Oy = sweepO(), Dy = cDy * own;
          XO[0] = fma(K.cc_oz[0], My, XO[0]); XD[0] = fma(K.cc_oz[2], My, XD[0]); XM[1] = fma(K.cc_mx[0], Oy, fma(K.cc_mx[2], Dy, XM[1])); }
      }
      __syncthreads();
#pragma unroll
      for (int c = 0; c < 3; ++c) Lw[(c * TYR + r) * TXN + lx] = cD * xc[c];
      __syncthreads();
      if (!halo_wave) {
        { nb(0); const double My = sweepM(); XO[2] = fma(K.cc_oz[2], My, XO[2]); XD[2] = fma(K.cc_oz[3], My, XD[2]); }
        { nb(1); const double Oy = sweepO(), Dy = cDy * own; XM[2] = fma(K.cc_mx[2], Oy, fma(K.cc_mx[3], Dy, XM[2])); }
        { nb(2); const double My = sweepM(), Oy = sweepO(), Dy = cDy * own;
          XO[0] = fma(K.cc_oz[1], My, XO[0]); XD[0] = fma(K.cc_oz[3], My, XD[0]); XM[1] = fma(K.cc_mx[1], Oy, fma(K.cc_mx[3], Dy, XM[1])); }
      }
      __syncthreads();
      if (!halo_wave) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double v = xstage(XK[c], XM[c], XO[c], XD[c]);
          if constexpr (CHEB == 2) { double *sd = kk == 0 ? a.cheb.side_lo : a.cheb.side_hi; if (sd && out) sd[3 * (j * NX + i) + c] = v; }   // see kron_tile
          emit(c, v);
        }
      }
    }
    // rotate the window by one plane; the prefetched plane gets its Dirichlet columns zeroed as it enters
#pragma unroll
    for (int c = 0; c < 3; ++c) { const double t0 = W0[c]; W0[c] = W1[c]; W1[c] = W2[c]; W2[c] = t0; }
    apply_mask(W2, mp);
  }
  if (a.dot_partials) {
    if (!out) dot_acc = 0.0;                                   // lanes without an output accumulated finite garbage
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_xor(dot_acc, off, 64);
    __syncthreads();
    if (lane == 0) L[w] = dot_acc;
    __syncthreads();
    if (tid == 0) { double t = 0; for (int q = 0; q < 16; ++q) t += L[q]; a.dot_partials[blockIdx.x] = t; }
  }
}

__global__ void __launch_bounds__(1024)
k_kron3_q1(KronArgs a, const double *__restrict__ x, double *__restrict__ y) {
  extern __shared__ double L[];
  if (a.pcg && (a.pcg->done || a.pcg->finishing)) return;
  int col, zc; tile_of(a, col, zc);
  if (col < a.n64 * a.nty64) kron_tile_q1<64, 0>(a, x, y, L, 62 * (col / a.nty64) - 1, col % a.nty64, zc);
  else kron_tile_q1<32, 0>(a, x, y, L, a.x0_32, col - a.n64 * a.nty64, zc);
}
__global__ void __launch_bounds__(1024)
k_kron3_q1_cheb(KronArgs a, const double *__restrict__ x, double *__restrict__ y) {
  extern __shared__ double L[];
  if (a.pcg && (a.pcg->done || a.pcg->finishing)) return;
  int col, zc; tile_of(a, col, zc);
  if (col < a.n64 * a.nty64) kron_tile_q1<64, 1>(a, x, y, L, 62 * (col / a.nty64) - 1, col % a.nty64, zc);
  else kron_tile_q1<32, 1>(a, x, y, L, a.x0_32, col - a.n64 * a.nty64, zc);
}
__global__ void __launch_bounds__(1024)
k_kron3_q1_cheb_side(KronArgs a, const double *__restrict__ x, double *__restrict__ y) {
  extern __shared__ double L[];
  if (a.pcg && (a.pcg->done || a.pcg->finishing)) return;
  int col, zc; tile_of(a, col, zc);
  if (col < a.n64 * a.nty64) kron_tile_q1<64, 2>(a, x, y, L, 62 * (col / a.nty64) - 1, col % a.nty64, zc);
  else kron_tile_q1<32, 2>(a, x, y, L, a.x0_32, col - a.n64 * a.nty64, zc);
}

// ---- 2D (Q2 and Q1): A_xx = (l+2G) Kx (x) My + G Mx (x) Ky,  A_yy = G Kx (x) My + (l+2G) Mx (x) Ky,
//      A_xy = l Cx (x) Cy^T + G Cx^T (x) Cy,  A_yx = A_xy^T.   No march: one 64 x 16 tile per trip, the two components of the tile go through
// LDS for the y-stage (one row per wave), the x-stage is the same scatter-by-DPP as in 3D.  A workgroup loops over tiles when the fused
// dot product limits the grid to the number of partial slots.
template <int KU, bool CHEB> __global__ void __launch_bounds__(1024)
k_kron2(KronArgs a, const double *__restrict__ x, double *__restrict__ y) {
  constexpr int H = KU == 2 ? 2 : 1, VX = 64 - 2 * H, VY = 16 - 2 * H;
  __shared__ double L[2 * 16 * 64];
  __shared__ double sdot[16];
  if (a.pcg && (a.pcg->done || a.pcg->finishing)) return;
  const int tid = threadIdx.x, r = __builtin_amdgcn_readfirstlane(tid >> 6), lx = tid & 63;
  const int NX = a.nn[0], NY = a.nn[1];
  const int ntx = (NX + VX - 1) / VX, ntiles = a.nblocks;
  const KronConsts &K = a.k;
  const bool halo_wave = r < H || r > 15 - H;
  double dot_acc = 0.0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int X0 = VX * (tile % ntx) - H, Y0 = VY * (tile / ntx) - H;
    const int i = X0 + lx, j = Y0 + r;
    const bool vn = i >= 0 && i < NX && j >= 0 && j < NY;
    const bool out = vn && lx >= H && lx <= 63 - H && !halo_wave;
    double u[2] = {0.0, 0.0};
    const int64_t node = (int64_t)j * NX + i;
    if (vn) {
      u[0] = x[node * 2]; u[1] = x[node * 2 + 1];
      if (a.constrained && (a.mask_anywhere || i == 0 || i == NX - 1 || j == 0 || j == NY - 1)) { const unsigned m = a.nodemask[node]; if (m & 1u) u[0] = 0.0; if (m & 2u) u[1] = 0.0; }
    }
    L[r * 64 + lx] = u[0]; L[(16 + r) * 64 + lx] = u[1];
    __syncthreads();
    if (!halo_wave) {
      const double mLx = i > 0 ? 1.0 : 0.0, mRx = i < NX - 1 ? 1.0 : 0.0, mLy = j > 0 ? 1.0 : 0.0, mRy = j < NY - 1 ? 1.0 : 0.0;
      // centre coefficients and band forms of the integer 1D matrices (see the 3D kernels): vertex / mid parity only exists for Q2
      const bool even_i = (lx & 1) == 0, odd_row = KU == 2 && (r & 1) != 0;
      double cMx, cKx, cDx, cMy, cKy, cDy, pe = 0.0;
      if constexpr (KU == 2) {
        pe = even_i ? 1.0 : 0.0;
        cMx = even_i ? 4.0 * (mLx + mRx) : 16.0; cKx = even_i ? 7.0 * (mLx + mRx) : 16.0; cDx = even_i ? 3.0 * (mLx - mRx) : 0.0;
        cMy = odd_row ? 16.0 : 4.0 * (mLy + mRy); cKy = odd_row ? 16.0 : 7.0 * (mLy + mRy); cDy = odd_row ? 0.0 : 3.0 * (mLy - mRy);
      } else { cMx = 2.0 * (mLx + mRx); cKx = mLx + mRx; cDx = mLx - mRx; cMy = 2.0 * (mLy + mRy); cKy = mLy + mRy; cDy = mLy - mRy; }
      double My[2], Ky[2], Oy[2], Dy[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const double *col = L + (c * 16 + r) * 64 + lx;
        const double nm1 = col[-64], np1 = col[64], own = col[0], s1 = nm1 + np1, d1 = nm1 - np1;
        if constexpr (KU == 2) {
          double s2 = 0.0, d2 = 0.0;
          if (!odd_row) { const double nm2 = col[-128], np2 = col[128]; s2 = nm2 + np2; d2 = np2 - nm2; }
          My[c] = fma(2.0, s1, cMy * own) - s2; Ky[c] = fma(-8.0, s1, cKy * own) + s2; Oy[c] = fma(4.0, d1, d2);
        } else { My[c] = fma(cMy, own, s1); Ky[c] = fma(cKy, own, -s1); Oy[c] = d1; }
        Dy[c] = cDy * own;
      }
      auto xstage = [&](const double FK, const double FM, const double FO, const double FD) {
        if constexpr (KU == 2) {
          const double t1 = fma(-8.0, FK, 2.0 * FM), t2 = pe * (FK - FM), pO = pe * FO;
          double sacc = fma(cKx, FK, fma(cMx, FM, cDx * FD));
          sacc += wave_up1(fma(4.0, FO, t1)) + wave_dn1(fma(-4.0, FO, t1));
          sacc += wave_up1(wave_up1(t2 - pO)) + wave_dn1(wave_dn1(t2 + pO));
          return sacc;
        } else {
          const double t1 = FM - FK;
          return fma(cKx, FK, fma(cMx, FM, cDx * FD)) + wave_up1(t1 + FO) + wave_dn1(t1 - FO);
        }
      };
      // K.xk_* = (l+2G | G) sKx sMy, K.m_lKyMz / m_gKyMz = (l+2G | G) sMx sKy, K.cc_mz = {-(l+G), l-G, G-l, l+G} sC^2
      const double yx = xstage(K.xk_l2g * My[0], K.m_gKyMz * Ky[0], fma(K.cc_mz[1], Dy[1], K.cc_mz[0] * Oy[1]), fma(K.cc_mz[3], Dy[1], K.cc_mz[2] * Oy[1]));
      const double yy = xstage(K.xk_g * My[1], K.m_lKyMz * Ky[1], fma(K.cc_mz[2], Dy[0], K.cc_mz[0] * Oy[0]), fma(K.cc_mz[3], Dy[0], K.cc_mz[1] * Oy[0]));
      if (out) {
        if constexpr (CHEB) {   // the Chebyshev root-form update where the product leaves the registers (see kron_tile): z_{j+1} = z_j + omega_j D^-1 (g - A z_j), g . z_{j+1}
          const double *ctab = a.cheb.tab + 2u * a.cheb.cls[node];
          const double g0 = a.cheb.g[node * 2], g1 = a.cheb.g[node * 2 + 1];
          const double z0 = fma(a.cheb.omega * ctab[0], g0 - yx, u[0]), z1 = fma(a.cheb.omega * ctab[1], g1 - yy, u[1]);
          a.cheb.znew[node * 2] = z0; a.cheb.znew[node * 2 + 1] = z1; dot_acc = fma(g0, z0, fma(g1, z1, dot_acc));
        } else { y[node * 2] = yx; y[node * 2 + 1] = yy; dot_acc = fma(u[0], yx, fma(u[1], yy, dot_acc)); }
      }
    }
    if (tile + (int)gridDim.x < ntiles) __syncthreads();
  }
  if (a.dot_partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot_acc += __shfl_xor(dot_acc, off, 64);
    if (lx == 0) sdot[r] = dot_acc;
    __syncthreads();
    if (tid == 0) { double t = 0; for (int q = 0; q < 16; ++q) t += sdot[q]; a.dot_partials[blockIdx.x] = t; }
  }
}

// y_i = diag_i x_i on the Dirichlet rows (ConstraintMatrix elimination), from the constraint list
__global__ void __launch_bounds__(256)
k_kron_fix_constrained(int64_t n, const int32_t *__restrict__ dofs, const double *__restrict__ diag_local, const double *__restrict__ x, double *__restrict__ y,
                       double *dot_partials /* optional: slot per block for sum x_i y_i over the Dirichlet rows */) {
  __shared__ double sh[4];
  double acc = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t d = dofs[t]; const double xv = x[d], yv = diag_local[d] * xv; y[d] = yv; acc = fma(xv, yv, acc);
  }
  if (dot_partials) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) dot_partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  }
}

// checks the integer element matrices above against Gauss quadrature of the Lagrange basis (4 points on [0,1], exact to degree 7)
void check_q2_element_matrices() {
  const double gx[4] = {0.0694318442029737124, 0.3300094782075718676, 0.6699905217924281324, 0.9305681557970262876};
  const double gw[4] = {0.1739274225687269287, 0.3260725774312730713, 0.3260725774312730713, 0.1739274225687269287};
  const double Mi[3][3] = {{4, 2, -1}, {2, 16, 2}, {-1, 2, 4}}, Ki[3][3] = {{7, -8, 1}, {-8, 16, -8}, {1, -8, 7}}, Ci[3][3] = {{-3, -4, 1}, {4, 0, -4}, {-1, 4, 3}};
  double M[3][3] = {{0}}, K[3][3] = {{0}}, C[3][3] = {{0}};
  for (int q = 0; q < 4; ++q) {
    const double t = gx[q];
    const double v[3] = {2 * (t - 0.5) * (t - 1), 4 * t * (1 - t), 2 * t * (t - 0.5)}, d[3] = {4 * t - 3, 4 - 8 * t, 4 * t - 1};
    for (int i = 0; i < 3; ++i) for (int jj = 0; jj < 3; ++jj) { M[i][jj] += gw[q] * v[i] * v[jj]; K[i][jj] += gw[q] * d[i] * d[jj]; C[i][jj] += gw[q] * d[i] * v[jj]; }
  }
  for (int i = 0; i < 3; ++i) for (int jj = 0; jj < 3; ++jj)
    if (std::fabs(30 * M[i][jj] - Mi[i][jj]) > 1e-12 || std::fabs(3 * K[i][jj] - Ki[i][jj]) > 1e-12 || std::fabs(6 * C[i][jj] - Ci[i][jj]) > 1e-12)
      throw Error("Q2 1D element matrices do not match their integer form");
}

}  // namespace

bool kron_supported(int dim, int k_u) { return (dim == 2 || dim == 3) && (k_u == 1 || k_u == 2); }

void kron_prepare_device() {
  const int lds = (int)((size_t)NFLD * kTileNodes * sizeof(double));
  PORO_HIP(hipFuncSetAttribute((const void *)k_kron3_q2, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  PORO_HIP(hipFuncSetAttribute((const void *)k_kron3_q1, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  PORO_HIP(hipFuncSetAttribute((const void *)k_kron3_q2_cheb, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  PORO_HIP(hipFuncSetAttribute((const void *)k_kron3_q1_cheb, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  PORO_HIP(hipFuncSetAttribute((const void *)k_kron3_q2_cheb_side, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  PORO_HIP(hipFuncSetAttribute((const void *)k_kron3_q1_cheb_side, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
}

int kron_apply(hipStream_t s, const MfArgs &m, const double *x, double *y, bool constrained, int n_cus, double *dot_partials, hipEvent_t ev0, hipEvent_t ev1, const PcgScalars *pcg,
               const KronCheb *cheb) {
  static bool checked = false;   // a host-side arithmetic check of the integer 1D matrices (no device state involved: once per process is right for any number of devices)
  if (!checked) { check_q2_element_matrices(); checked = true; }
  KronArgs a{};
  const int ku = m.k_u;
  for (int d = 0; d < 3; ++d) a.nn[d] = d < m.dim ? ku * m.box.n[d] + 1 : 1;
  if (m.dim == 2) {
    const int H = ku == 2 ? 2 : 1, VX = 64 - 2 * H, VY = 16 - 2 * H;
    const int64_t ntiles = (int64_t)((a.nn[0] + VX - 1) / VX) * ((a.nn[1] + VY - 1) / VY);
    a.nblocks = (int)ntiles;
    const double lam = m.lam, G = m.G, l2g = lam + 2 * G, c[4] = {-(lam + G), lam - G, G - lam, lam + G};
    const double mdiv = ku == 2 ? 30.0 : 6.0, kmul = ku == 2 ? 1.0 / 3.0 : 1.0, sC = ku == 2 ? 1.0 / 6 : 0.5;
    const double sMx = m.box.h[0] / mdiv, sMy = m.box.h[1] / mdiv, sKx = kmul / m.box.h[0], sKy = kmul / m.box.h[1];
    a.k.xk_l2g = l2g * sKx * sMy; a.k.xk_g = G * sKx * sMy; a.k.m_lKyMz = l2g * sMx * sKy; a.k.m_gKyMz = G * sMx * sKy;
    for (int i = 0; i < 4; ++i) a.k.cc_mz[i] = c[i] * sC * sC;
    a.nodemask = m.nodemask; a.constrained = constrained ? 1 : 0; a.mask_anywhere = m.mask_anywhere; a.dot_partials = dot_partials; a.pcg = pcg;
    const unsigned grid = (unsigned)(dot_partials ? std::min<int64_t>(ntiles, kMaxPartials) : ntiles);
    if (cheb) {
      a.cheb = *cheb;
      if (ku == 2) hipExtLaunchKernelGGL((k_kron2<2, true>), dim3(grid), dim3(1024), 0, s, ev0, ev1, 0, a, x, y);
      else hipExtLaunchKernelGGL((k_kron2<1, true>), dim3(grid), dim3(1024), 0, s, ev0, ev1, 0, a, x, y);
    } else if (ku == 2) hipExtLaunchKernelGGL((k_kron2<2, false>), dim3(grid), dim3(1024), 0, s, ev0, ev1, 0, a, x, y);
    else hipExtLaunchKernelGGL((k_kron2<1, false>), dim3(grid), dim3(1024), 0, s, ev0, ev1, 0, a, x, y);
    return (int)grid;
  }
  // tile shapes: (64 lanes x 16 rows) and (32 x 32); valid outputs 60 x 12 / 28 x 28 for Q2 (halo 2), 62 x 14 / 30 x 30 for Q1 (halo 1)
  const int vx64 = ku == 2 ? 60 : 62, vx32 = ku == 2 ? 28 : 30, vy64 = ku == 2 ? 12 : 14, vy32 = ku == 2 ? 28 : 30, halo = ku == 2 ? 2 : 1;
  // x-extent = full 64-lane tiles + (when what is left fits) one 32-lane tile column
  a.n64 = a.nn[0] / vx64; const int rem = a.nn[0] - vx64 * a.n64;
  a.has32 = 0;
  if (rem > vx32) a.n64 += 1; else if (rem > 0) a.has32 = 1;
  a.x0_32 = vx64 * a.n64 - halo;
  a.nty64 = (a.nn[1] + vy64 - 1) / vy64; a.nty32 = (a.nn[1] + vy32 - 1) / vy32;
  // z-chunks: as many workgroups as fit the chip in ONE round (one 1024-thread workgroup per CU), an even number of planes each
  const int cols = a.n64 * a.nty64 + a.has32 * a.nty32;
  // (Q2 planes come in vertex / mid pairs), balanced to within one unit; chunks of fewer than 8 planes would be mostly halo
  a.zunit = ku == 2 ? 2 : 1;
  const int units = a.nn[2] / a.zunit;
  static const int min_chunk_env = std::getenv("PORO_KRON_MIN_CHUNK") ? std::atoi(std::getenv("PORO_KRON_MIN_CHUNK")) : 0;
  // shortest chunk: 8 planes (less is mostly halo) - unless the box is so thin (a slab of a partitioned run) that 8-plane chunks leave most CUs without a workgroup:
  // then 4-plane chunks, which cost more loads per output but halve the length of the march every workgroup has to finish
  int min_chunk = min_chunk_env > 0 ? min_chunk_env : 8;
  if (min_chunk_env <= 0 && (int64_t)cols * ((units * a.zunit + 7) / 8) < n_cus) min_chunk = 4;       // (72 x 72 x 9 cells, one slab of 8: 23.0 -> 18.6 us per application; x 18: 27.3 -> 23.8)
  const int upc = std::max(1, min_chunk / a.zunit);              // units in a shortest chunk
  int nzc = n_cus / cols; if (nzc > (units + upc - 1) / upc) nzc = (units + upc - 1) / upc; if (nzc < 1) nzc = 1;
  // 32-bit buffer offsets are relative to a chunk's first plane: a chunk (its planes + the halo + one prefetched pair) must span less than 2^31 bytes
  { const int64_t plane_bytes = (int64_t)a.nn[0] * a.nn[1] * 24, max_planes = (((int64_t)1 << 31) - 1) / plane_bytes - 2 * halo - 3;
    if (max_planes < 2 * a.zunit) throw Error("structured operator: an x-y plane of the box is too large for 32-bit buffer offsets");
    const int need = (int)((a.nn[2] + max_planes - 1) / max_planes);
    if (nzc < need) nzc = std::min(need, units); }
  a.nzc = nzc; a.zq = units / nzc; a.zr = units % nzc;
  a.nA = a.n64 * a.nty64 * a.nzc; a.nblocks = a.nA + a.has32 * a.nty32 * a.nzc; a.cols = cols;
  { static const int zm = std::getenv("PORO_KRON_COLMAJOR") ? 0 : 1; a.zmajor = zm; }
  const double lam = m.lam, G = m.G, l2g = lam + 2 * G, c[4] = {-(lam + G), lam - G, G - lam, lam + G};
  const double mdiv = ku == 2 ? 30.0 : 6.0, kmul = ku == 2 ? 1.0 / 3.0 : 1.0, sC = ku == 2 ? 1.0 / 6 : 0.5;   // scales of the integer 1D matrices
  const double sM[3] = {m.box.h[0] / mdiv, m.box.h[1] / mdiv, m.box.h[2] / mdiv}, sK[3] = {kmul / m.box.h[0], kmul / m.box.h[1], kmul / m.box.h[2]};
  KronConsts &k = a.k;
  k.xk_l2g = l2g * sM[1] * sK[0] * sM[2]; k.xk_g = G * sM[1] * sK[0] * sM[2];
  k.m_gKyMz = G * sK[1] * sM[0] * sM[2]; k.m_gMyKz = G * sM[1] * sM[0] * sK[2]; k.m_lKyMz = l2g * sK[1] * sM[0] * sM[2]; k.m_lMyKz = l2g * sM[1] * sM[0] * sK[2];
  for (int i = 0; i < 4; ++i) { k.cc_mz[i] = c[i] * sC * sC * sM[2]; k.cc_oz[i] = c[i] * sM[1] * sC * sC; k.cc_mx[i] = c[i] * sC * sM[0] * sC; }
  a.nodemask = m.nodemask; a.constrained = constrained ? 1 : 0; a.mask_anywhere = m.mask_anywhere;
  const size_t lds = (size_t)NFLD * kTileNodes * sizeof(double);
  const int nblk = a.nblocks;
  // one partial slot per workgroup: a box whose x-y extent needs more workgroups than slots runs without the fused x.y (negative return
  // value = "no partials written"; the caller launches its separate dot kernel, exactly as for the assembled operator)
  const bool fuse = dot_partials && nblk <= kMaxPartials / 2;
  a.dot_partials = fuse ? dot_partials : nullptr; a.pcg = pcg;
  // ev0 / ev1 (optional): timestamps at the start / end of THIS dispatch, so the measured time is the kernel's own duration
  if (cheb) {
    a.cheb = *cheb;
    const bool side = cheb->side_lo || cheb->side_hi;
    if (ku == 2) { if (side) hipExtLaunchKernelGGL(k_kron3_q2_cheb_side, dim3((unsigned)nblk), dim3(1024), lds, s, ev0, ev1, 0, a, x, y); else hipExtLaunchKernelGGL(k_kron3_q2_cheb, dim3((unsigned)nblk), dim3(1024), lds, s, ev0, ev1, 0, a, x, y); }
    else { if (side) hipExtLaunchKernelGGL(k_kron3_q1_cheb_side, dim3((unsigned)nblk), dim3(1024), lds, s, ev0, ev1, 0, a, x, y); else hipExtLaunchKernelGGL(k_kron3_q1_cheb, dim3((unsigned)nblk), dim3(1024), lds, s, ev0, ev1, 0, a, x, y); }
  } else if (ku == 2) hipExtLaunchKernelGGL(k_kron3_q2, dim3((unsigned)nblk), dim3(1024), lds, s, ev0, ev1, 0, a, x, y);
  else hipExtLaunchKernelGGL(k_kron3_q1, dim3((unsigned)nblk), dim3(1024), lds, s, ev0, ev1, 0, a, x, y);
  return (dot_partials && !fuse) ? -nblk : nblk;
}

// Dirichlet rows of y = A_c x after kron_apply (separate launch so that profiles attribute it separately); slot_base = first free
// partial slot (the return value of kron_apply) when the dot product is fused
void kron_fix_constrained(hipStream_t s, const MfArgs &m, const double *x, double *y, double *dot_partials, int slot_base) {
  if (!m.n_dirichlet) return;
  // one Dirichlet dof per thread when the partial slots allow it (the list kernel is latency-bound)
  const int nfix = (int)std::min<int64_t>((m.n_dirichlet + 255) / 256, dot_partials ? std::max(1, kMaxPartials - slot_base) : 4096);
  hipLaunchKernelGGL(k_kron_fix_constrained, (unsigned)nfix, 256, 0, s, m.n_dirichlet, m.dirichlet_dofs, m.diag_local, x, y, dot_partials ? dot_partials + slot_base : (double *)nullptr);
}

}  // namespace poro
