// Shared declarations of the HIP back end (gfx950 only): device buffers, the context behind the
// opaque poro_ctx handle and the launch wrappers of every kernel family.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <type_traits>
#include <stdexcept>
#include <string>
#include <list>
#include <vector>
#include "../../include/poroel_hip.h"

namespace poro {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

#define PORO_HIP(x)                                                                                   \
  do { hipError_t e_ = (x); if (e_ != hipSuccess) throw poro::Error(std::string(#x) + " -> " + hipGetErrorString(e_)); } while (0)

template <class T> struct DevBuf {
  T *p = nullptr; size_t n = 0;
  DevBuf() = default; DevBuf(const DevBuf &) = delete; DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { release(); }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
  void alloc(size_t n_) { release(); n = n_; if (n) PORO_HIP(hipMalloc((void **)&p, n * sizeof(T))); }
  void zero(hipStream_t s) { if (n) PORO_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
  void upload(const T *h, size_t n_) { alloc(n_); if (n) PORO_HIP(hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice)); }
  void upload(const std::vector<T> &h) { upload(h.data(), h.size()); }
};

constexpr int kMaxPartials = 1024;   // grid cap of every reducing kernel = number of per-block partial sums
constexpr int kScalarSlots = 32;

// device-resident scalars of a PCG solve (no host round trip inside the iteration)
struct PcgScalars {
  double dh, gg, gz;         // reductions of the current iteration (after all-reduce)
  double gh2[2];             // g.z of the previous iteration (ping-pong by iteration parity)
  double alpha, beta;
  double tol, res0, res;
  int32_t it, done, converged, max_iter, finishing, pad_;
};

// Mailbox in pinned, device-visible host memory: a one-block kernel at the end of an enqueued sequence copies a few scalars (and optionally the PCG state) into it and
// then raises `seq` with a system-scope release; the host spins on `seq` instead of issuing a device-to-host copy + stream synchronisation (which idled the GPU
// for 20-60 us per round trip, 25 times per time step: profiles/r02_gpu_idle_gaps.txt).
struct Mailbox { volatile unsigned long long seq; unsigned long long pad_; double vals[16]; PcgScalars sc; };

// device-side state of the single-reduction PCG (partitioned runs)
struct Cg1State { double gamma_old, alpha_old, alpha, beta, tol, res0, res; int32_t it, done, converged, pad_; };

// INVERSE Jacobi diagonal handed to the PCG kernels (they multiply): the full vector of reciprocals, or (uniform boxes) a class byte
// per node + table[class][component] of reciprocals
// reciprocal Jacobi diagonal; a ZERO entry marks an inert (Dirichlet) dof that PCG leaves alone; `inert` is the same set as a byte mask
struct DiagVec { const double *full = nullptr; const uint8_t *cls = nullptr; const double *tab = nullptr; int ncomp = 1; const uint8_t *inert = nullptr;
                 const double *z = nullptr; /* explicit preconditioner: z = P^-1 g is supplied as a vector (pcg() fills it between the two update kernels) */
                 double *z1_out = nullptr; double z1_scale = 0; /* optional: the residual update also stores z1_scale * D^-1 g (first Chebyshev iterate) */ };

struct FeTablesDev {   // device copies of poro_fe_tables
  int nq_u, nq_p, nq_f, ns_u, ns_p;
  const double *w_qu, *w_qp, *w_qf, *u_qu, *du_qu, *du_qp, *q1_qu, *dq1_qu, *q1_qp, *dq1_qp, *u_qf, *dq1_qf;
};

// uniform-box coupling operator (kernels_box.hip): 1D local blocks N[s][t] = int phi_s psi_t, D[s][t] = int phi_s' psi_t on the unit interval
struct BoxCoupling { int k; int n[3]; double h[3]; double N[3][2], D[3][2]; };

// fast diagonalisation (kernels_fdm.hip): per direction the generalised eigenvectors S (n x n row-major, columns M-orthonormal), S^T, eigenvalues
struct FdmDir { int n = 0; DevBuf<double> S, St, lam; };
struct FdmScalar { int dim = 0; FdmDir dir[3]; bool built = false; };
struct FdmScale { const double *lam[3]; int n[3]; double a, k[3]; int64_t ncol, col0, col_total; };   // divide by a + k0 lam0[i] + k1 lam1[j] + k2 lam2[k] at grid node (i, j, k); ncol > 0: column-distributed layout
// partitioned (slab) form: the transforms of the leading directions are local, the last direction runs on columns gathered by an all-to-all
struct FdmWindow { int n_planes; int64_t ncols_valid, grid_col0, grid_plane0; };   // one peer's share of a window copy (fdm_window_batch)
struct FdmDist {
  bool built = false; int n_ranks = 1, rank = 0;
  std::vector<int> layers, off;                 // cell layers and first global plane of every rank
  int ng = 0;                                   // global node planes of the last direction
  int64_t ncol_total = 0, C = 0;                // columns (nodes of one plane) and columns per rank
  int max_own = 0, max_nl = 0;                  // padded plane counts of the two exchanges
  FdmDir last;                                  // global eigenvectors of the last direction
  DevBuf<double> sendbuf, recvbuf, tz1, tz2; std::vector<double> hsend, hrecv;
  DevBuf<FdmWindow> windows;                    // [4][n_ranks]: the per-peer windows of the four copies around the two all-to-alls (one launch each)
};

// block fast diagonalisation of the displacement system (kernels_fdmu.hip): per (component, direction) the transform matrices S^T (fwd) and
// S (bwd) in MFMA fragment order and the eigenvalues (inf marks removed modes); coef[c][d] = lambda + 2G (d == c) | G
struct FdmuDir { int n = 0; bool reg_form = false, split = false, blk = false; int n_even = 0; int blk_kk[2] = {0, 0}, blk_nch[2] = {0, 0}, blk_mb[2] = {0, 0} /* [forward, backward] */; DevBuf<double> fwd, bwd, lam; };
struct FdmU { int dim = 0; int nn[3] = {1, 1, 1}; double coef[3][3] = {}; FdmuDir dir[3][3]; FdmuDir last_global[3]; bool built = false, single = false;
              int fix[3][3][2] = {};
              // slab-partitioned form: node planes of the last direction are gathered per column group by an all-to-all (as FdmDist for the Q1 systems)
              bool dist = false; int n_ranks = 1, rank = 0; std::vector<int> layers, off /* first global node plane of every rank */; int ng = 0; int64_t ncol_total = 0, C = 0;
              int max_own = 0, max_nl = 0; DevBuf<double> sendbuf, recvbuf, tz1, tz2; };
// octant form of the block fast diagonalisation (kernels_fdmo.hip; 3D boxes, one rank, every direction mirror-symmetric for every component).
// The even / odd butterflies of the three directions commute with everything inside the preconditioner, so they are hoisted out of it: the CG residual g and
// the preconditioned residual z live as 8 octants per component, Q[c][o][kz][ky][kx] with o = 4 pz + 2 py + px (p = 0: even part e_k = v_k + v_k', 1: odd
// part o_k = v_k - v_k' of the pair k < h, k' = n - 1 - k; the centre node of an odd line is its own mirror: e = v, o = 0), h = (n + 1) / 2 entries per
// direction.  Each of the 24 (component, octant) blocks is then an independent half-size 3D transform, done in three passes (x y fused per plane, z
// forward + eigenvalue scaling + z backward, y x fused per plane); the butterflies ride in the CG update kernels, which touch every entry anyway.
struct FdmOct {
  bool built = false; int nt = 0;                 // 16-wide MFMA tiles per half line (max over the directions), 1..8
  int n[3] = {1, 1, 1}, h[3] = {1, 1, 1}; int hxp = 2;   // hxp: row pitch = h[0] rounded up to even (16-byte aligned rows; the pad entry is zero and stays zero)
  int64_t co_stride = 0, n_oct = 0;               // entries per (component, octant) = hxp h[1] h[2]; 24 co_stride
  double coef[3][3] = {};
  DevBuf<double> fwd[3][3][2], bwd[3][3][2], lam[3][3][2];   // [component][direction][parity]: half-size transforms in MFMA fragment order [tile][4 nt][64], eigenvalues (inf = no such mode)
  DevBuf<double> g, z, t;                         // residual, preconditioned residual, scratch - all in octant form
  std::vector<double> h_lam[3][3][2];             // host copies of the eigenvalues
  DevBuf<double> bxy;                             // [component][py][px][my][mx] = coef_x lam_x[mx] + coef_y lam_y[my]: the part of the eigenvalue sum a z line shares (pass 2 reads it per column)
  // slab-partitioned form ("quadrant form"): only x and y are split into parities in the CG vectors, Q[c][q = 2 py + px][kz local][ky][kx] (no = 4 blocks per component);
  // the z transform runs on whole global lines after an all-to-all of column chunks, where the z butterfly is applied on the way in / out of the transposed array
  int no = 8;                                     // blocks per component in g, z, t: 8 octants, or 4 quadrants when z is not split locally
  int nc = 3;                                     // components per node: 3; 2 for the planar (2D) form, which is the quadrant form with a single plane
  bool planar = false;                            // 2D: the transforms are batched tiled GEMMs over whole (component, quadrant) planes (fwd = row-major h x h matrices F[mode][node])
  int own_z = 0;                                  // local node planes that count in dot products (the upper shared plane belongs to the neighbour)
  struct Slab {
    bool on = false; int n_ranks = 1, rank = 0;
    int ng = 0, hzg = 0;                          // global nodes of a z line / rows of a transposed block (half length with the parity split, ng without)
    bool zboth = false; DevBuf<int64_t> dst2;      // z stage with both parity parts of a line per workgroup (no transposed array, no scatter kernel): [ng][2] target offsets of every global plane
    int np = 2, nb = 12;                          // parity parts of the z direction (2; scalar Q1 systems: 1 = no butterfly), blocks of the local arrays (3 components x 4 quadrants; scalar: 1)
    int cw = 64, nchunk = 0, cps = 0, chunk0 = 0, my_chunks = 0;   // chunk width (columns), chunks per (component, quadrant) plane, chunks per rank share, this rank's first global chunk and count
    int64_t scols = 0;                            // columns of a share = cps cw
    int own = 0, nl = 0, max_own = 0, max_nl = 0, rows_back = 0;    // planes this rank sends (owns) / holds; maxima over the ranks; sum of the ranks' local planes
    int64_t recv_off = 0;                         // buf = [send | recv], each [rank][plane][scols]; a rank's own block is written straight into the receive half
    DevBuf<int64_t> row_in;                       // [ng]: offset (in buf) of global plane kz after the gathering all-to-all
    DevBuf<int64_t> row_out; DevBuf<int32_t> row_kz;   // [rows_back]: offset (in buf) of a row of the scattering all-to-all, global plane of that row
    DevBuf<double> buf, tz;                       // all-to-all buffers; transposed array [chunk][pz][hzg][cw]
  } slab;
  struct ScalarTable { double a, kappa; DevBuf<double> t; };
  std::list<ScalarTable> scalar_tables;           // scalar form: a + kappa (lam_x + lam_y) per plane position, one table per (a, kappa) seen (pressure Jacobian, mass matrix)
};
// dependency levels of the lower / upper triangle in natural row order (rows of one level can be swept concurrently)
struct SsorLevels { DevBuf<int32_t> fwd_rows, bwd_rows; std::vector<int64_t> fwd_off, bwd_off; bool built = false; };
struct CsrDev {
  int64_t n = 0, nnz = 0;
  DevBuf<int64_t> rp; DevBuf<int32_t> col; DevBuf<int64_t> diag_pos;
  int lanes_per_row = 8;
  SsorLevels ssor;
};

// closed affine constraints (hanging nodes) on the device: forward lists (constrained dof -> masters) for expand / distribute and the
// transposed lists (master -> constrained dofs) so that the reduction C^T y is a gather without atomics
struct ConsDev {
  int64_t n = 0, n_masters = 0;
  DevBuf<int32_t> dof, master; DevBuf<int64_t> ptr; DevBuf<double> weight, inhom;
  DevBuf<int32_t> t_master, t_dof; DevBuf<int64_t> t_ptr; DevBuf<double> t_weight;
  DevBuf<uint8_t> inert;   // byte mask: constrained (Dirichlet or hanging) dofs stay out of the Krylov system
  bool any_inhom = false;
};

struct Timer { double seconds = 0; int64_t launches = 0 /* with events */, enqueued = 0 /* all */; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
               bool sample(int stride) { return enqueued++ % stride == 0; } };

// interface of a general partition in one dof space (poro_partition.shared_*): concatenated per-neighbour lists for packing / receiving, and per
// shared dof the sources of its sum in ascending rank order (-1 = this rank's own partial value, otherwise a position in `recv`)
struct IfcDev {
  int64_t m_send = 0, m_shared = 0, n_owned = 0;
  std::vector<int64_t> ptr;                 // [n_neighbours + 1] into dof / send / recv
  DevBuf<int32_t> dof, sh_dof, sh_src; DevBuf<int64_t> sh_ptr; DevBuf<double> send, recv;
  std::vector<double> hsend, hrecv;         // host staging of the callback communicator
};

struct Comm {
  poro_partition part{};
  bool general = false; std::vector<int32_t> neighbours; IfcDev ifc_u, ifc_p;
  // RCCL (resolved at run time from librccl.so.1)
  void *nccl_comm = nullptr;
  // host-staged callbacks (tests)
  poro_allreduce_fn ar = nullptr; poro_sendrecv_fn sr = nullptr; void *user = nullptr;
  DevBuf<double> recv_lo, recv_hi; std::vector<double> hsend, hrecv;
  bool force_multi = false;
  bool multi() const { return part.n_ranks > 1 || force_multi; }
};

struct BoxDev { int enabled = 0; int n[3] = {1, 1, 1}; int nn[3] = {1, 1, 1}; double h[3] = {1, 1, 1}; };

}  // namespace poro

struct poro_ctx {
  poro_ctx() = default; poro_ctx(const poro_ctx &) = delete; poro_ctx &operator=(const poro_ctx &) = delete;
  ~poro_ctx() { if (mailbox) (void)hipHostFree(mailbox); }
  int device = 0, operator_mode = PORO_OP_CSR;
  hipStream_t stream = nullptr;
  int dim = 2, k_u = 2, ns_u = 9, ns_p = 4, dpc_u = 18, dpc_p = 4, nv = 4;
  int64_t n_cells = 0, n_u = 0, n_p = 0;
  poro_material mat{};
  poro::BoxDev box;
  // the logical tensor structure the fast-diagonalisation preconditioners work on: the uniform box, or a tensor-product grid without the box tag (poro_desc.tensor)
  struct Lines { bool on = false, uniform = true; int n[3] = {1, 1, 1}, nn[3] = {1, 1, 1}; std::vector<double> hcell[3]; } lines;
  poro::Comm comm;
  // mesh / dof data
  poro::DevBuf<int32_t> cell_dofs_u, cell_dofs_p, color_cells;
  poro::DevBuf<double> cell_X, tables;
  poro::DevBuf<double> cell_geo;          // 3D meshes whose cells are all parallelepipeds: J^-1 (9, row-major [b][d] = d xi_b / d x_d) and det J per cell (kernels_mfg.hip)
  poro::FeTablesDev fe{};
  std::vector<int64_t> color_off;            // host offsets into color_cells
  poro::DevBuf<uint8_t> dir_mask, node_mask; poro::DevBuf<double> dir_val; poro::DevBuf<int32_t> dir_dofs;
  std::vector<int32_t> h_dir_dof; std::vector<double> h_dir_val;
  poro::ConsDev cons_u, cons_p;
  poro::DevBuf<uint8_t> pdir_mask; poro::DevBuf<double> pdir_val; int64_t n_pdir = 0;   // extension: prescribed pressures
  poro::DevBuf<int32_t> bface_cell, bface_local, bface_id, neu_label, neu_comp; poro::DevBuf<double> neu_val;
  int64_t n_bfaces = 0; int n_neumann = 0;
  // matrices
  poro::CsrDev Ap, Au;                       // pressure pattern (mass, Laplace, Jacobian share it), displacement pattern
  poro::DevBuf<double> Mp, Kp, Jp, Au_val;
  bool projection_matrix_ready = false;
  poro::DevBuf<double> Ke;                   // reference element matrix of the matrix-free operator
  // vectors (ids of include/poroel_hip.h)
  std::map<int, poro::DevBuf<double>> vec, vec_saved;
  poro::DevBuf<double> lift_u, neumann_u, diag_u, diag_u_local, diag_J, diag_M, src_local;
  poro::DevBuf<double> dinv_u, dinv_J, dinv_M;   // reciprocals of the Jacobi diagonals
  poro::DevBuf<uint8_t> diag_u_cls; poro::DevBuf<double> diag_u_tab;   // dictionary form of diag_u (uniform boxes): class per node + table[class][dim]
  poro::DevBuf<double> wg_u, wd_u, wh_u, wg_p, wd_p, wh_p, tmp_p, proj_y;
  bool box_asm_checked = false;
  int box_asm = 0 /* 0 off, 1 unchecked, 2 checked against the per-cell kernels */; poro::BoxCoupling box_cpl{};
  poro::DevBuf<double> ilu_u, ilu_J, ilu_M; bool ilu_u_valid = false, ilu_J_valid = false, ilu_M_valid = false;   // ILU(0) factors on the CSR patterns
  poro::DevBuf<double> wz_p;   // z = P^-1 g of an explicit preconditioner (pressure-sized systems)
  poro::FdmScalar fdm_p; poro::FdmDist fdm_dist; poro::DevBuf<double> fdm_t1, fdm_t2;   // fast diagonalisation of the Q1 box operators
  double cheb_lmax = 0;   // estimate of lambda_max(D^-1 A_u) (Lanczos at matrix build; 0 = not yet computed)
  double cheb_ratio_default = 0;   // default interval ratio of the Chebyshev preconditioner, from the GLOBAL mesh size (0 = not yet computed)
  poro::DevBuf<double> cheb_z, cheb_t;
  poro::FdmOct fdm_oct;
  poro::FdmOct fdm_p_fused;          // the scalar Q1 systems through the same transform kernel (3D boxes, lines of <= 128 vertices)
  // two-level preconditioner (poro_desc.coarse): the underlying uniform box as a context of its own (same device and stream) + the node-wise interpolation P and its transpose
  struct Interp { int64_t n_fine = 0, n_coarse = 0; int lanes = 1, lanes_t = 1;   /* lanes per row of the interpolation kernels, from the mean row length */ poro::DevBuf<int64_t> p_ptr, pt_ptr; poro::DevBuf<int32_t> p_col, pt_col; poro::DevBuf<double> p_w, pt_w; };   // P (rows = fine) and its transpose as CSR
  struct TwoLevel : Interp { poro_ctx *box = nullptr; Interp pressure; } two_level;   // (the base part: displacement nodes; .pressure: pressure dofs, optional)
  bool borrowed_stream = false;     // (the box context of a two-level preconditioner runs on its parent's stream)
  poro::FdmU fdm_u; poro::DevBuf<double> fdmu_t1, fdmu_t2, wz_u; int fdm_u_state = 0 /* 0 unknown, 1 usable, -1 not separable */; std::string fdm_u_why;
  std::vector<uint8_t> h_node_mask;
  poro::DevBuf<double> partials; poro::DevBuf<poro::PcgScalars> scal; poro::DevBuf<double> red;   // red: kScalarSlots doubles
  poro::Mailbox *mailbox = nullptr; unsigned long long mb_seq = 0;   // pinned host memory (hipHostMalloc), device-visible at the same address
  bool cheb_z1_ready = false;   // single-reduction PCG: the update kernel has already stored the first Chebyshev iterate of the coming preconditioner call
  int timing_stride = 1;   // events on every timing_stride-th launch of a family (poro_timers_enable)
  poro::DevBuf<double> cheb_side_lo, cheb_side_hi;   // partial products of the fused Chebyshev kernel on the shared planes (slab partitions)
  poro::DevBuf<poro::Cg1State> cg1_state; poro::DevBuf<double> cg1_z[2], cg1_w[2];   // single-reduction PCG of partitioned runs ([0]: displacement-sized, [1]: pressure-sized)
  bool matrix_built = false;
  int interleaved_u = 0;
  int pcg_hint_fdm_u[2] = {0, 0}, pcg_hint_cheb_u[2] = {0, 0}; int64_t cheb_applies = 0;
  int pcg_hint_u[2] = {0, 0};   // iterations of the last two displacement solves (batch scheduling of the next one)
  int pcg_hint_p[2] = {0, 0}, pcg_hint_proj[2] = {0, 0};   // the same for the iterative pressure / projection solves
  int n_cus = 256; int mf_variant = 1 /* 0 element-matrix gather, 1 sum-factorised (where supported) */; int mask_anywhere = 0;
  // timing
  bool timing = false; std::map<std::string, poro::Timer> timers; std::vector<hipEvent_t> event_pool;
  double jac_dt = -1;
};

namespace poro {

// ---- kernels_la.hip -----------------------------------------------------------------------------
void la_fill(hipStream_t s, double *x, double v, int64_t n);
// copy n <= 16 doubles from `src` (device) and optionally *sc into the host mailbox, then publish sequence number `seq` (system-scope release)
// node-wise sparse interpolation of a node-interleaved vector: out[(row, c)] = sum_k w[k] in[(col[k], c)] (prolongation by P, restriction by its transpose)
void la_nodal_interp(hipStream_t s, const int64_t *ptr, const int32_t *col, const double *w, int64_t n_rows, int ncomp, const double *in, double *out, int lanes = 1);
// z = omega D^-1 g + P z_c, zero on the inert dofs (additive two-level preconditioner)
void la_two_level_combine(hipStream_t s, const int64_t *ptr, const int32_t *col, const double *w, int64_t n_rows, int ncomp, const double *zc, const double *g, const double *dinv, const uint8_t *inert, double omega, double *z, int lanes = 1);
void la_copy_many(hipStream_t s, int count, double *const *dst, const double *const *src, const int64_t *n);   // several device-to-device copies in one launch
void la_post(hipStream_t s, Mailbox *mb, unsigned long long seq, const double *src, int n, const PcgScalars *sc);
void la_copy(hipStream_t s, double *y, const double *x, int64_t n);
void la_axpy(hipStream_t s, double *y, double a, const double *x, int64_t n);
void la_add_range(hipStream_t s, double *y, const double *x, int64_t n);
void la_add_two_ranges(hipStream_t s, double *y0, const double *x0, double *y1, const double *x1, int64_t n);
// partials-based reductions; results land in red[slot..] after la_reduce_finish
void la_dot_partials(hipStream_t s, const double *a, const double *b, int64_t n, double *partials /*[kMaxPartials]*/, const PcgScalars *gate = nullptr /* no-op once gate->done / finishing */);
void la_norm_partials(hipStream_t s, const double *a, int64_t n, double *partials_l2, double *partials_inf);
void la_reduce_finish(hipStream_t s, const double *partials, int n_sets, double *red /*[n_sets]*/, int max_not_sum_mask);
void la_csr_spmv(hipStream_t s, const CsrDev &A, const double *val, const double *x, double *y);
// R = -(M t + kappa K p + src)
void la_csr_residual(hipStream_t s, const CsrDev &A, const double *M, const double *K, double kappa, const double *t, const double *p,
                     const double *src, double *R);
void la_pressure_tmp(hipStream_t s, double *t, const double *ev, const double *ev0, const double *p, const double *p_old, double c1, double c2, int64_t n);
void la_jacobian(hipStream_t s, double *J, const double *M, const double *K, double a, double kappa, int64_t nnz);
void la_ifc_pack(hipStream_t s, const IfcDev &I, const double *v);
void la_ifc_sum(hipStream_t s, const IfcDev &I, double *v);
void la_csr_diag(hipStream_t s, const CsrDev &A, const double *val, double *diag);
void la_reciprocal(hipStream_t s, double *y, const double *x, int64_t n);
void la_pointwise_mul(hipStream_t s, double *y, const double *x, int64_t n);   // y *= x
// polynomial preconditioner in root form around the (inverse) Jacobi diagonal: z1 = s D^-1 g;  z_new = z_j + omega D^-1 (g - A z_j);
// gz_partials != nullptr: block partials of g . z_new over the first n_owned entries
void la_cheb_first(hipStream_t s, double *z, const double *g, const DiagVec &dv, double scale, int64_t n);
void la_cheb_step(hipStream_t s, double *znew, const double *zj, const double *g, const double *Az, const DiagVec &dv, double omega, int64_t n, int64_t n_owned, double *gz_partials);
// shared planes of a slab partition after the fused kernel: z_{j+1} = z_j + omega D^-1 (g - (own partial + neighbour's partial)) on the first (lo) / last (hi) `plane` dofs of a vector of n
void la_cheb_fix_planes(hipStream_t s, double *znew, const double *zj, const double *g, const double *own_lo, const double *nbr_lo, const double *own_hi, const double *nbr_hi, const DiagVec &dv, double omega, int64_t n, int64_t plane);
// x[dof_i] = sum_k w_k x[master_k] (+ inhomogeneity_i): ConstraintMatrix::distribute; with_inhom = false inside the Krylov iteration
void la_cons_expand(hipStream_t s, const ConsDev &C, double *x, bool with_inhom);
// y <- C^T y: y[master] += sum w y[dof_i], then y[dof_i] = 0 (ConstraintMatrix::condense of a vector)
void la_cons_reduce(hipStream_t s, const ConsDev &C, double *y);
void la_xpby(hipStream_t s, double *y, double a, double b, const double *x, int64_t n);   // y = a y + b x
void la_ilu0_factor(hipStream_t s, const CsrDev &A, const SsorLevels &lv, double *lu /* in: A's values, out: L (unit diagonal) and U */, int *flag /* device int, 0 = ok, else 1 + row of a zero pivot */);
void la_ilu_apply(hipStream_t s, const CsrDev &A, const double *lu, const SsorLevels &lv, const double *src, double *dst);
void la_ssor_apply(hipStream_t s, const CsrDev &A, const double *val, const SsorLevels &lv, double omega, const double *src, double *dst);
void la_sum_strains(hipStream_t s, double *ev, const double *const *strains, int n, int64_t len);
void la_effective_stress(hipStream_t s, const double *const *strains, double *const *stresses, int dim, double lam, double G, int64_t len);
void la_set_constrained(hipStream_t s, double *x, const uint8_t *mask, const double *val, int64_t n);
void la_rhs_u_finish(hipStream_t s, double *rhs, const double *lift, const double *neumann, const uint8_t *mask, int64_t n);
// PCG pieces (device-side control, see solver in ctx.hip)
void pcg_init_residual(hipStream_t s, double *g, const double *Ax, const double *b, const uint8_t *inert /*nullable*/, int64_t n);
void la_mask_zero(hipStream_t s, double *x, const uint8_t *mask, int64_t n);
void pcg_dot_dh(hipStream_t s, const PcgScalars *sc, const double *d, const double *h, int64_t n_owned, double *partials);
void pcg_first_direction(hipStream_t s, double *d, const double *g, const DiagVec &diag, int prec, int64_t n, int64_t n_owned, double *partials /*2 sets: gg, gz*/);
void pcg_scalars_sum(hipStream_t s, const double *partials, int n_sets, double *red);
void cg1_dots(hipStream_t s, const double *g, const double *z, const double *w, const double *b /*nullable: adds b.b*/, int64_t n_owned, double *partials /*4 sets*/);
void cg1_scalars(hipStream_t s, Cg1State *st, const double *red, int first, double abs_tol, double rel_tol, int max_iter, int stop_rule);
void cg1_update(hipStream_t s, const Cg1State *st, double *d, double *sv, double *x, double *g, const double *z, const double *w, const DiagVec &dv /* inert mask; z1_out: also store z1_scale D^-1 g_new */, int64_t n);
void pcg_scalars_start(hipStream_t s, PcgScalars *sc, const double *red /*bb, gg, gz*/, double abs_tol, double rel_tol, int max_iter, int stop_rule);
// single-rank fast path: the consumers reduce the block partials themselves (no scalar kernels, no host round trip);
// parity = iteration index & 1 selects the g.z slot read / written
void pcg_update_g_fused(hipStream_t s, PcgScalars *sc, int parity, double *g, const double *h, const DiagVec &diag, int prec, int64_t n, int64_t n_owned,
                        const double *partials_dh, const double *red /*null: single rank*/, double *partials_out /*2 sets*/);
void pcg_update_d_fused(hipStream_t s, PcgScalars *sc, int parity, int it, double *x, double *d, const double *g, const DiagVec &diag, int prec, int64_t n,
                        const double *partials_in /*2 sets*/, const double *red /*null: single rank*/);

// ---- kernels_asm.hip ----------------------------------------------------------------------------
struct AsmArgs {
  int dim, k_u, ns_u, ns_p, nv, dpc_u;
  FeTablesDev fe;
  const int32_t *cell_dofs_u, *cell_dofs_p; const double *cell_X;
  const double *cell_geo;   // [n_cells][10] or null (see poro_ctx::cell_geo)
  const uint8_t *dir_mask; const double *dir_val;
  poro_material mat;
  int interleaved_u;   // dof = node * dim + component everywhere (lets K-asm-u look CSR positions up per node pair)
};
void asm_u_matrix(hipStream_t s, const AsmArgs &a, const int32_t *cells, int64_t n_cells, const int64_t *rp, const int32_t *col, double *val, double *lift);
void asm_u_element_matrix(hipStream_t s, const AsmArgs &a, int32_t cell, double *Ke);
void asm_u_rhs(hipStream_t s, const AsmArgs &a, const int32_t *cells, int64_t n_cells, const double *p, double *rhs);
void asm_u_neumann(hipStream_t s, const AsmArgs &a, int64_t n_bfaces, const int32_t *bf_cell, const int32_t *bf_local, const int32_t *bf_id,
                   int n_neu, const int32_t *label, const int32_t *comp, const double *value, double *rhs);
void asm_p_matrices(hipStream_t s, const AsmArgs &a, const int32_t *cells, int64_t n_cells, const int64_t *rp, const int32_t *col, double *M, double *K, double *src);
void asm_proj_rhs(hipStream_t s, const AsmArgs &a, const int32_t *cells, int64_t n_cells, const double *u, int n_comp, const int32_t *comps /*host*/,
                  double *const *rhs /*host array of device ptrs*/);

// ---- kernels_mfg.hip: matrix-free operator on general meshes (one wave per cell, coloured scatter) ----------------
void mfg_apply(hipStream_t s, const AsmArgs &a, const int32_t *color_cells, const std::vector<int64_t> &color_off, int64_t n_u, const double *x, double *y, bool constrained, int mode);
// ---- kernels_mf.hip -----------------------------------------------------------------------------
struct MfArgs { int dim, k_u; BoxDev box; const double *Ke; const uint8_t *mask; const double *diag_local; double lam, G; int mask_anywhere; const uint8_t *nodemask; const int32_t *dirichlet_dofs; int64_t n_dirichlet; };
void mf_apply(hipStream_t s, const MfArgs &a, const double *x, double *y, bool constrained, double *dot_partials = nullptr);
void mf_diag(hipStream_t s, const MfArgs &a, double *diag);
// y = (a M + kappa K) x for the Q1 pressure space of a uniform box (constant-coefficient 3^dim-point stencil)
void p_stencil_apply(hipStream_t s, int dim, const BoxDev &box, double a, double kappa, const double *x, double *y);
void p_residual_stencil(hipStream_t s, int dim, const BoxDev &box, double kappa, const double *t, const double *p, const double *src, double *R);

// polynomial-preconditioner step fused into the structured operator's stores: z_new = z_j + omega D^-1 (g - A z_j), D^-1 in dictionary form; z_new != z_j
struct KronCheb { const double *g = nullptr; double *znew = nullptr; double omega = 0; const uint8_t *cls = nullptr; const double *tab = nullptr;
                  // slab partitions: the raw partial product A z_j on the first / last node plane (the planes shared with the lower / upper neighbour) is written here as well
                  double *side_lo = nullptr, *side_hi = nullptr; };
// ---- kernels_kron.hip: sum-factorised (Kronecker) form of the same operator ---------------------------
bool kron_supported(int dim, int k_u);
void kron_prepare_device();   // per-device function attributes (dynamic LDS opt-in) of the structured kernels; call after hipSetDevice
BoxCoupling box_coupling(int dim, int k_u, const BoxDev &box);
void box_rhs_u(hipStream_t s, int dim, const BoxCoupling &B, double alpha, const double *p, const double *lift, const double *neu, const uint8_t *mask, double *rhs);
void box_asm_u_matrix(hipStream_t s, int dim, int k_u, const BoxDev &box, const double *Ke, const CsrDev &A, const uint8_t *mask, double *val);
void box_proj_rhs(hipStream_t s, int dim, const BoxCoupling &B, const double *u, int n_comp, const int32_t *tensor_components, double *const *rhs);
void q1_eig(int n_cells, double h, std::vector<double> &S, std::vector<double> &lam);   // host: generalised eigenpairs of the 1D Q1 stiffness / mass matrices
// all peers' windows in one launch: block q of the dense side is dense + q blk (to_block: peer `self` goes to dense_self instead - its own block of the receive buffer)
void fdm_window_batch(hipStream_t s, double *grid, double *dense, double *dense_self, int self, bool to_block, const FdmWindow *win, int n_peers, int n_planes_pad, int64_t C, int64_t grid_stride, int64_t blk);
void fdm_window(hipStream_t s, double *dst, const double *src, bool to_block, int n_planes, int n_planes_pad, int64_t C, int64_t ncols_valid, int64_t grid_stride, int64_t grid_col0, int64_t grid_plane0);
void fdm_transform(hipStream_t s, const double *T, int n_l, int64_t SI, int64_t n_outer, const double *in, double *out, const FdmScale *scale);
void fdm_apply(hipStream_t s, const FdmScalar &F, double a, const double k[3], const double *g, double *z, double *t1, double *t2);
// ---- kernels_fdmu.hip ---------------------------------------------------------------------------
double jacobi_scaled_lambda_max(int n, const std::vector<double> &A);
double sym_lambda_max(int n, const std::vector<double> &A);   // largest eigenvalue of a small dense symmetric matrix
void fdmu_eig_1d(int k, int n_cells, double h, bool fix_lo, bool fix_hi, std::vector<double> &S, std::vector<double> &lam);
void fdmu_eig_1d(int k, const std::vector<double> &cell_sizes, bool fix_lo, bool fix_hi, std::vector<double> &S, std::vector<double> &lam);   // the same on a non-uniform 1D grid
void fdmu_upload_dir(FdmuDir &D, const std::vector<double> &S, const std::vector<double> &lam, int nn, bool single, bool allow_split);   // D.split tells whether the even / odd form was taken
// stage 2: the whole application (single rank); 0 / 1: the passes of the leading directions before / after the caller's distributed last direction
void fdmu_apply(hipStream_t s, const FdmU &F, const double *g, double *z, void *t1, void *t2, int stage);
void fdmu_window(hipStream_t s, double *dst, const double *src, bool to_block, int ncomp, int n_planes, int n_planes_pad, int64_t C, int64_t ncols_valid,
                 int64_t grid_stride, int64_t grid_planes, int64_t grid_col0, int64_t grid_plane0);
void fdmu_lines(hipStream_t s, const FdmU &F, const FdmuDir *last_dir, int64_t C, int64_t col0, int64_t ncol_valid, void *in, void *out);
// ---- kernels_fdmo.hip: octant form (see FdmOct) --------------------------------------------------
bool fdmo_usable(int dim, const int nn[3]);          // 3D, half lines of at most 128 entries (8 MFMA tiles)
void fdmo_init(FdmOct &O, const int nn[3], const double coef[3][3], hipStream_t s);   // sizes + buffers
// slab-partitioned form: nn = LOCAL nodes; layers[q] = node planes rank q holds minus one (its cell layers x degree); the last direction's matrices are uploaded for the GLOBAL line
void fdmo_init_slab(FdmOct &O, const int nn[3], const double coef[3][3], int rank, const std::vector<int> &node_layers, bool has_upper, hipStream_t s);
bool fdmo_upload_dir(FdmOct &O, int comp, int dir, const std::vector<double> &S, const std::vector<double> &lam, int nn);
void fdmo_finalize(FdmOct &O);   // after every (component, direction) has been uploaded: derived tables
void fdmo_apply(hipStream_t s, const FdmOct &O, const double *g_oct, double *z_oct, double *scratch_oct, const PcgScalars *gate = nullptr, hipEvent_t *ev /* optional: 3 start / stop pairs attached to the three pass dispatches */ = nullptr);   // z = blockdiag(A_cc)^-1 g, all in octant form; gate: no-op once gate->done / finishing
// the same transform kernel for the scalar Q1 systems of a 3D box (nodal layout, one block set, no octants): 3 launches instead of 6
bool fdmo_scalar_usable(int dim, const int nn[3]);
void fdmo_scalar_init(FdmOct &O, const int nn[3], hipStream_t s);
void fdmo_scalar_upload_dir(FdmOct &O, int dir, const std::vector<double> &S, const std::vector<double> &lam, int n);
void fdmo_scalar_apply(hipStream_t s, FdmOct &O, double a, double kappa, const double *g, double *z, const PcgScalars *gate = nullptr);
void fdmo_scalar_apply_many(hipStream_t s, FdmOct &O, double a, double kappa, int nb, const double *const *g, double *const *z, const PcgScalars *gate = nullptr);   // up to 3 right-hand sides in one set of launches
// block partials of |y_e - b_e|^2 and |b_e|^2 for up to three (y, b) pairs: sets 2e and 2e + 1 of `partials`
void la_residual_norms_many(hipStream_t s, int nb, const double *const *y, const double *const *b, int64_t n, double *partials);
// slab partitions: nn = LOCAL vertices, the last direction's matrices are uploaded for the GLOBAL line; the three sweeps separately (all-to-alls in between, ctx_prec.hip)
void fdmo_scalar_init_slab(FdmOct &O, const int nn[3], int rank, const std::vector<int> &node_layers, hipStream_t s);
void fdmo_scalar_slab_pass(hipStream_t s, FdmOct &O, int pass, double a, double kappa, const double *in, double *out);
// planar (2D) form: quadrant layout with one plane and 2 components; four batched GEMM launches per application
bool fdmo_planar_usable(int dim, const int nn[3]);
void fdmo_init_planar(FdmOct &O, const int nn[3], const double coef[3][3], hipStream_t s, bool split = true /* false: different conditions at the two ends of some line - no parity split, full-length transforms */);
bool fdmo_upload_dir_planar(FdmOct &O, int comp, int dir, const std::vector<double> &S, const std::vector<double> &lam, int nn);
void fdmo_apply_planar(hipStream_t s, const FdmOct &O, const double *g_q, double *z_q, const PcgScalars *gate = nullptr);
void fdmo_from_nodal(hipStream_t s, const FdmOct &O, const double *v_nodal, double *q_oct);   // q = H v (node-interleaved vector -> octant form)
void fdmo_to_nodal(hipStream_t s, const FdmOct &O, const double *r_oct, double *v_nodal);     // v = H^-1-form of the backward transform: v_k = a + b, v_k' = a - b
// the vector kernels of pcg() with g / z in octant form (same device-side scalar protocol as their nodal counterparts in kernels_la.hip)
void fdmo_init_residual(hipStream_t s, const FdmOct &O, double *g_oct, const double *Ax, const double *b, const uint8_t *inert);
void fdmo_first_direction(hipStream_t s, const FdmOct &O, double *d, const double *g_oct, const double *z_oct, double *partials /*2 sets: gg, gz*/);
// red != null (partitioned runs): the all-reduced d.h (update_g: red[0]) / g.g and g.z (update_d: red[0], red[1]) instead of the block partials
void fdmo_update_g(hipStream_t s, const FdmOct &O, PcgScalars *sc, int parity, double *g_oct, const double *h, const uint8_t *inert, const double *partials_dh, double *partials_out /*gg*/, const double *red = nullptr);
void fdmo_update_d(hipStream_t s, const FdmOct &O, PcgScalars *sc, int parity, int it, double *x, double *d, const double *z_oct, const double *partials_in /*2 sets*/, const double *red = nullptr);
void fdmo_dot_owned(hipStream_t s, const FdmOct &O, const double *a_oct, const double *b_oct, double *partials, const PcgScalars *gate);   // block partials of a.b over the planes this rank owns
// slab form: the pieces of one application around the two all-to-alls (ctx_prec.hip drives them)
void fdmo_slab_pass(hipStream_t s, const FdmOct &O, int pass /*1, 2, 3*/, const double *in, double *out, const PcgScalars *gate, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);
// pass 1: quadrant layout -> slab.buf (send half; own share -> receive half); pass 2: gathered planes in slab.buf -> slab.tz; pass 3: scattered planes in slab.buf -> quadrant layout
void fdmo_slab_scatter_pack(hipStream_t s, const FdmOct &O, const PcgScalars *gate);                            // slab.tz -> slab.buf (inverse butterfly; every rank's planes incl. the shared ones)
// dot_partials (optional, kMaxPartials slots, zeroed by the caller once): per-workgroup partial sums of x.y, fused into the apply
int kron_apply(hipStream_t s, const MfArgs &a, const double *x, double *y, bool constrained, int n_cus, double *dot_partials = nullptr, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
               const PcgScalars *pcg = nullptr /* launch becomes a no-op once pcg->done / finishing is set */,
               const KronCheb *cheb = nullptr /* 3D only: store the Chebyshev update instead of the product (y is not written) */);   // returns the workgroup count (= partial slots used)
void kron_fix_constrained(hipStream_t s, const MfArgs &a, const double *x, double *y, double *dot_partials, int slot_base);

}  // namespace poro
