// rmh_comm.hip -- pooling the chains of several GPUs for diagnostics: one RCCL
// all-gather of the ranks' state blocks and one all-reduce of their moment
// accumulators over xGMI.  Not in the sampling loop.
// The reference has nothing distributed (README.md:21-23).
//
// librccl is opened with dlopen at the first call, so that the sampling entry
// points carry no link-time dependency on it.
#include <algorithm>
#include "../../include/riemann_hip.h"
#include "rmh_internal.hpp"
#include "rmh_mfma.hpp"

#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

namespace rmh {
int comm_fail(int code, const std::string &msg); // rmh_api.hip
hipStream_t comm_stream();
int comm_state_info(rmh_state_t s, double **theta, int64_t *N, int64_t *ld, int *d, int *dp);
int comm_hist_info(rmh_history_t h, rmh_state_t *s, double **thetas, int64_t *entries, int64_t *rows);
} // namespace rmh
using namespace rmh;

namespace {

typedef struct { char internal[128]; } nccl_uid_t;
typedef void *nccl_comm_t;
enum { kNcclFloat64 = 8, kNcclSum = 0 }; // rccl.h: ncclDataType_t / ncclRedOp_t

struct Rccl {
    void *so = nullptr;
    int (*GetUniqueId)(nccl_uid_t *) = nullptr;
    int (*CommInitRank)(nccl_comm_t *, int, nccl_uid_t, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, nccl_comm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(nccl_comm_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
} g_rccl;
nccl_comm_t g_comm = nullptr;
int g_nranks = 0, g_rank = 0;
hipEvent_t g_cev0 = nullptr, g_cev1 = nullptr; // around the last all-gather
// Device staging for the small host collectives, allocated with the communicator: a rank whose
// hipMalloc failed right before a collective would stay out of it and leave the others hanging in
// it.  Where a call still has to allocate, the ranks first agree on a status word
// (comm_agree) so that all of them enter the data collective or none does.
constexpr size_t kStageDoubles = 1 << 17;      // 1 MiB
double *g_stage = nullptr, *g_flag = nullptr;
double g_last_ms = 0.0;
int64_t g_last_bytes = 0;

int load_rccl()
{
    if (g_rccl.so) return RMH_OK;
    void *so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!so) so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!so) so = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!so) return comm_fail(RMH_ERR_HIP, std::string("cannot open librccl: ") + dlerror());
#define SYM(field, name)                                                              \
    *reinterpret_cast<void **>(&g_rccl.field) = dlsym(so, name);                      \
    if (!g_rccl.field) return comm_fail(RMH_ERR_HIP, std::string("librccl lacks ") + name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(AllGather, "ncclAllGather");
    SYM(AllReduce, "ncclAllReduce");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.so = so;
    return RMH_OK;
}

#define RMH_NCCL(expr)                                                                        \
    do {                                                                                      \
        int r_ = (expr);                                                                      \
        if (r_ != 0)                                                                          \
            return comm_fail(RMH_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

// per-chain-block partial moments: out[0] += n, out[1+p] += sum_c theta[p][c],
// out[1+d+p*d+q] += sum_c theta[p][c] theta[q][c]  -- one workgroup per (p, q)
__global__ void k_moments(const double *theta, int64_t N, int64_t ld, int d, double *out)
{
    __shared__ double red[4];
    const int p = blockIdx.x / d, q = blockIdx.x % d;
    double acc = 0.0, accp = 0.0;
    for (int64_t c = threadIdx.x; c < N; c += blockDim.x) {
        const double a = theta[(int64_t)p * ld + c];
        acc += a * theta[(int64_t)q * ld + c];
        accp += a;
    }
    for (int pass = 0; pass < 2; ++pass) {
        double v = pass ? accp : acc;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double t = red[0] + red[1] + red[2] + red[3];
            if (pass == 0) out[1 + d + p * d + q] = t;
            else if (q == 0) out[1 + p] = t;
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (double)N;
}

} // namespace

namespace rmh {
int comm_active() { return g_comm != nullptr; }
// in-place sum over the ranks of a device buffer, on the library's stream
int comm_allreduce_sum(double *dev, size_t count)
{
    if (!g_comm) return comm_fail(RMH_ERR_STATE, "no communicator (rmh_comm_init)");
    const int r = g_rccl.AllReduce(dev, dev, count, kNcclFloat64, kNcclSum, g_comm, comm_stream());
    if (r != 0) return comm_fail(RMH_ERR_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
    return RMH_OK;
}
} // namespace rmh

extern "C" {

int rmh_comm_unique_id(uint8_t *id128)
{
    int rc = load_rccl();
    if (rc) return rc;
    nccl_uid_t id;
    RMH_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, 128);
    return RMH_OK;
}

int rmh_comm_init(const uint8_t *id128, int nranks, int rank)
{
    int rc = rmh_init(-1); // the device must be bound already
    if (rc) return rc;
    if ((rc = load_rccl())) return rc;
    if (g_comm) return comm_fail(RMH_ERR_STATE, "rmh_comm_init: communicator exists");
    if (nranks < 1 || rank < 0 || rank >= nranks) return comm_fail(RMH_ERR_PARAM, "rmh_comm_init: bad rank");
    nccl_uid_t id;
    memcpy(id.internal, id128, 128);
    if (!g_stage) {
        hipError_t e = hipMalloc(&g_stage, (kStageDoubles + 1) * sizeof(double));
        if (e != hipSuccess) return comm_fail(RMH_ERR_HIP, hipGetErrorString(e));
        g_flag = g_stage + kStageDoubles;
    }
    RMH_NCCL(g_rccl.CommInitRank(&g_comm, nranks, id, rank));
    g_nranks = nranks;
    g_rank = rank;
    return RMH_OK;
}

int rmh_comm_destroy(void)
{
    if (g_comm) {
        RMH_NCCL(g_rccl.CommDestroy(g_comm));
        g_comm = nullptr;
    }
    return RMH_OK;
}

int rmh_comm_info(int *nranks, int *rank)
{
    if (nranks) *nranks = g_comm ? g_nranks : 0;
    if (rank) *rank = g_comm ? g_rank : 0;
    return RMH_OK;
}

// sum / max / min over the ranks of a small host buffer (timing and bookkeeping values)
// max over the ranks of a status word (0 = fine): every rank learns whether any of them failed
// locally, before a data collective that all of them must enter.  Uses the preallocated flag.
static int comm_agree(int local_err)
{
    hipStream_t st = comm_stream();
    double v = local_err ? 1.0 : 0.0;
    hipError_t e = hipMemcpyAsync(g_flag, &v, sizeof(double), hipMemcpyHostToDevice, st);
    int r = 0;
    if (e == hipSuccess) r = g_rccl.AllReduce(g_flag, g_flag, 1, kNcclFloat64, 2 /* ncclMax */, g_comm, st);
    if (e == hipSuccess && r == 0) e = hipMemcpyAsync(&v, g_flag, sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && r == 0) e = hipStreamSynchronize(st);
    if (r != 0) return comm_fail(RMH_ERR_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
    if (e != hipSuccess) return comm_fail(RMH_ERR_HIP, hipGetErrorString(e));
    if (v != 0.0)
        return comm_fail(RMH_ERR_HIP, local_err ? "a local allocation or launch failed before the collective"
                                                : "another rank failed before the collective");
    return RMH_OK;
}

int rmh_comm_allreduce_host(double *buf, int64_t n, int op)
{
    if (!g_comm) return comm_fail(RMH_ERR_STATE, "rmh_comm_allreduce_host: no communicator (rmh_comm_init)");
    if (!buf || n < 1 || op < 0 || op > 2) return comm_fail(RMH_ERR_PARAM, "rmh_comm_allreduce_host: bad argument");
    static const int kOp[3] = {0, 2, 3}; // rccl.h ncclRedOp_t: ncclSum, ncclMax, ncclMin
    hipStream_t st = comm_stream();
    // through the preallocated staging buffer, in pieces: no allocation between the ranks' entries
    for (int64_t o = 0; o < n; o += (int64_t)kStageDoubles) {
        const size_t m = (size_t)std::min<int64_t>(n - o, (int64_t)kStageDoubles);
        hipError_t e = hipMemcpyAsync(g_stage, buf + o, m * sizeof(double), hipMemcpyHostToDevice, st);
        int r = 0;
        if (e == hipSuccess) r = g_rccl.AllReduce(g_stage, g_stage, m, kNcclFloat64, kOp[op], g_comm, st);
        if (e == hipSuccess && r == 0) e = hipMemcpyAsync(buf + o, g_stage, m * sizeof(double), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && r == 0) e = hipStreamSynchronize(st);
        if (r != 0) return comm_fail(RMH_ERR_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
        if (e != hipSuccess) return comm_fail(RMH_ERR_HIP, hipGetErrorString(e));
    }
    return RMH_OK;
}

// every rank has drained its stream and reached this call
int rmh_comm_barrier(void)
{
    double one = 1.0;
    return rmh_comm_allreduce_host(&one, 1, 0);
}

// chains held by every rank (counts[nranks], may be null) and their sum
int rmh_comm_pooled_chains(rmh_state_t s, int64_t *total, int64_t *counts)
{
    if (!g_comm) return comm_fail(RMH_ERR_STATE, "rmh_comm_pooled_chains: no communicator (rmh_comm_init)");
    double *theta;
    int64_t N, ld;
    int d, dp;
    int rc = comm_state_info(s, &theta, &N, &ld, &d, &dp);
    if (rc) return rc;
    std::vector<double> cnt((size_t)g_nranks, 0.0);
    cnt[g_rank] = (double)N;
    if ((rc = rmh_comm_allreduce_host(cnt.data(), g_nranks, 0))) return rc;
    int64_t tot = 0;
    for (int k = 0; k < g_nranks; ++k) {
        if (counts) counts[k] = (int64_t)cnt[k];
        tot += (int64_t)cnt[k];
    }
    if (total) *total = tot;
    return RMH_OK;
}

int rmh_comm_last_ms(double *ms, int64_t *bytes_per_rank)
{
    if (ms) *ms = g_last_ms;
    if (bytes_per_rank) *bytes_per_rank = g_last_bytes;
    return RMH_OK;
}

// Shards may differ in size (shard_range hands out blocks that differ by one chain): every rank
// contributes a chain-major (Nmax, d) block, zero padded, to ONE ncclAllGather per block of states;
// the host copy keeps the first N_k chains of rank k.  `src` holds n_blk parameter-major (rows, ld)
// blocks `blk_stride` doubles apart (the current state: one; history entries: several);
// pooled: (n_blk, sum_k N_k, d).
static int allgather_blocks(const double *src, int64_t n_blk, int64_t blk_stride, int64_t N, int64_t ld, int d,
                            const std::vector<int64_t> &counts, double *pooled)
{
    int64_t nmax = 0, total = 0;
    for (int64_t n : counts) {
        nmax = n > nmax ? n : nmax;
        total += n;
    }
    hipStream_t st = comm_stream();
    const size_t block = (size_t)nmax * d;
    double *all = nullptr, *nd = nullptr;
    hipError_t e = hipMalloc(&all, block * g_nranks * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&nd, block * sizeof(double));
    if (e == hipSuccess && !g_cev0) e = hipEventCreate(&g_cev0);
    if (e == hipSuccess && !g_cev1) e = hipEventCreate(&g_cev1);
    // every rank enters the collectives below or none does
    int rc = comm_agree(e != hipSuccess);
    if (rc) {
        (void)hipFree(all); (void)hipFree(nd);
        return rc;
    }
    int r = 0;
    float ms = 0.f;
    for (int64_t b = 0; b < n_blk && e == hipSuccess && r == 0; ++b) {
        e = hipMemsetAsync(nd, 0, block * sizeof(double), st);
        if (e == hipSuccess) e = launch_transpose_dn_to_nd(src + b * blk_stride, nd, N, d, ld, 1, st);
        if (e == hipSuccess) e = hipEventRecord(g_cev0, st);
        // (a local failure from here on still enters the collective, with whatever the block holds,
        // and is reported afterwards: the other ranks must not be left waiting)
        r = g_rccl.AllGather(nd, all, block, kNcclFloat64, g_comm, st);
        if (e == hipSuccess && r == 0) e = hipEventRecord(g_cev1, st);
        int64_t off = 0;
        for (int k = 0; k < g_nranks && r == 0 && e == hipSuccess; ++k) {
            if (counts[k] > 0)
                e = hipMemcpyAsync(pooled + (b * total + off) * d, all + (size_t)k * block,
                                   (size_t)counts[k] * d * sizeof(double), hipMemcpyDeviceToHost, st);
            off += counts[k];
        }
        if (e == hipSuccess && r == 0) e = hipStreamSynchronize(st);
        if (e == hipSuccess && r == 0) {
            float f = 0.f;
            e = hipEventElapsedTime(&f, g_cev0, g_cev1);
            ms += f;
        }
    }
    if (e == hipSuccess && r == 0) {
        g_last_ms = ms;
        g_last_bytes = (int64_t)(block * sizeof(double)) * n_blk;
    }
    (void)hipFree(all); (void)hipFree(nd);
    if (r != 0) return comm_fail(RMH_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(r));
    if (e != hipSuccess) return comm_fail(RMH_ERR_HIP, hipGetErrorString(e));
    return RMH_OK;
}

// the ranks' chain counts, and a check that they all run one model (same d)
static int comm_counts(rmh_state_t s, int d, std::vector<int64_t> &counts)
{
    counts.assign((size_t)g_nranks, 0);
    int64_t total = 0;
    int rc = rmh_comm_pooled_chains(s, &total, counts.data());
    if (rc) return rc;
    double dd[2] = {(double)d, -(double)d};
    if ((rc = rmh_comm_allreduce_host(dd, 2, 1))) return rc;
    if (dd[0] != (double)d || dd[1] != -(double)d)
        return comm_fail(RMH_ERR_PARAM, "all-gather: the ranks' states differ in dimension");
    return RMH_OK;
}

int rmh_allgather_samples(rmh_state_t s, double *pooled)
{
    if (!g_comm) return comm_fail(RMH_ERR_STATE, "rmh_allgather_samples: no communicator (rmh_comm_init)");
    if (!pooled) return comm_fail(RMH_ERR_PARAM, "rmh_allgather_samples: null output");
    double *theta;
    int64_t N, ld;
    int d, dp;
    int rc = comm_state_info(s, &theta, &N, &ld, &d, &dp);
    if (rc) return rc;
    std::vector<int64_t> counts;
    if ((rc = comm_counts(s, d, counts))) return rc;
    return allgather_blocks(theta, 1, 0, N, ld, d, counts, pooled);
}

int rmh_allgather_history(rmh_history_t h, int64_t e0, int64_t e1, int64_t stride, double *pooled)
{
    if (!g_comm) return comm_fail(RMH_ERR_STATE, "rmh_allgather_history: no communicator (rmh_comm_init)");
    rmh_state_t s;
    double *thetas, *theta;
    int64_t entries, rows, N, ld;
    int d, dp;
    int rc = comm_hist_info(h, &s, &thetas, &entries, &rows);
    if (rc) return rc;
    if (!pooled || e0 < 0 || e1 > entries || e1 <= e0 || stride < 1)
        return comm_fail(RMH_ERR_PARAM, "rmh_allgather_history: bad entry range");
    if ((rc = comm_state_info(s, &theta, &N, &ld, &d, &dp))) return rc;
    std::vector<int64_t> counts;
    if ((rc = comm_counts(s, d, counts))) return rc;
    // all ranks must ask for the same entries: (e0, e1, stride) agree iff min == max of each
    double q[6] = {(double)e0, (double)e1, (double)stride, -(double)e0, -(double)e1, -(double)stride};
    if ((rc = rmh_comm_allreduce_host(q, 6, 1))) return rc;
    if (q[0] != (double)e0 || q[1] != (double)e1 || q[2] != (double)stride || q[3] != -(double)e0 ||
        q[4] != -(double)e1 || q[5] != -(double)stride)
        return comm_fail(RMH_ERR_PARAM, "rmh_allgather_history: the ranks ask for different entries");
    const int64_t n_sel = (e1 - e0 + stride - 1) / stride;
    return allgather_blocks(thetas + e0 * rows * ld, n_sel, stride * rows * ld, N, ld, d, counts, pooled);
}

int rmh_allreduce_moments(rmh_state_t s, double *moments)
{
    if (!g_comm) return comm_fail(RMH_ERR_STATE, "rmh_allreduce_moments: no communicator (rmh_comm_init)");
    double *theta;
    int64_t N, ld;
    int d, dp;
    int rc = comm_state_info(s, &theta, &N, &ld, &d, &dp);
    if (rc) return rc;
    hipStream_t st = comm_stream();
    const size_t n = 1 + (size_t)d + (size_t)d * d;
    double *acc = nullptr;
    hipError_t e = hipMalloc(&acc, n * sizeof(double));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_moments, dim3((unsigned)(d * d)), dim3(256), 0, st, theta, N, ld, d, acc);
        e = hipGetLastError();
    }
    if ((rc = comm_agree(e != hipSuccess))) { // every rank enters the all-reduce or none does
        (void)hipFree(acc);
        return rc;
    }
    int r = g_rccl.AllReduce(acc, acc, n, kNcclFloat64, kNcclSum, g_comm, st);
    if (e == hipSuccess && r == 0) e = hipMemcpyAsync(moments, acc, n * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess && r == 0) e = hipStreamSynchronize(st);
    (void)hipFree(acc);
    if (r != 0) return comm_fail(RMH_ERR_HIP, std::string("ncclAllReduce: ") + g_rccl.GetErrorString(r));
    if (e != hipSuccess) return comm_fail(RMH_ERR_HIP, hipGetErrorString(e));
    return RMH_OK;
}

} // extern "C"
