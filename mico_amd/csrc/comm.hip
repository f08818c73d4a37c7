// Collectives of the data-parallel alignment step over RCCL (xGMI), behind the C-ABI (include/mico_hip.h, "mico_comm_*").
// Replaces, for a binding that does not go through torch.distributed: concat_all_gather of data/utils/distributed.py:50-66 (one PACKED
// all-gather for every small per-step tensor instead of 3 + #subtasks latency-bound ones), the row exchange behind
// all_gather_with_grad(condition_feats)[neg_idx] (vast.py:421-433: index-then-fetch, see mico_amd/distributed.py), and DDP's gradient
// averaging (in-place all-reduce of arena slices / reduce-scatter + all-gather of flat buckets).
// RCCL is resolved at run time (dlopen of the librccl.so.1 already resident in the process - PyTorch-ROCm's own copy - or the ROCm one):
// a single-GPU user of libmico_hip.so needs no RCCL at all, and the library never links a second RCCL next to the framework's.
// Every call is asynchronous on the caller's stream, allocates nothing on the device (scratch is caller-owned) and returns 0 / negative codes.
#include "common.h"

#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace {

// the slice of the NCCL / RCCL API used here (rccl.h: stable C ABI, enum values as published)
typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
enum { NCCL_INT8 = 0, NCCL_UINT8 = 1, NCCL_FLOAT32 = 7 };
enum { NCCL_SUM = 0, NCCL_AVG = 4 };
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(nccl_uid*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
const char* g_rccl_err = nullptr;

void load_rccl() {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);       // the copy the host framework already loaded, if any
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) { g_rccl_err = "librccl.so.1 not found (dlopen)"; return; }
#define SYM(F) do { *(void**)(&g_rccl.F) = dlsym(h, "nccl" #F); if (!g_rccl.F) { g_rccl_err = "librccl lacks nccl" #F; return; } } while (0)
    SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(AllGather); SYM(AllReduce); SYM(ReduceScatter); SYM(Send); SYM(Recv);
    SYM(GroupStart); SYM(GroupEnd); SYM(GetErrorString);
#undef SYM
    g_rccl.handle = h;
}

struct Comm {
    nccl_comm c;
    int rank, nranks;
};

#define RCCL_OK(expr, what) do { const int rc_ = (expr); if (rc_ != 0) return mico_set_err(MICO_ELAUNCH, "%s: RCCL error %d (%s)", what, rc_, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc_) : "?"); } while (0)

constexpr int MAX_PARTS = 8;
struct PackArgs {
    const char* src[MAX_PARTS];
    int64_t row_bytes[MAX_PARTS], off[MAX_PARTS];
    int n;
    int64_t rows, total;
};

// out[r, off[p] + j] = src[p][r * row_bytes[p] + j]: the per-rank send buffer of the packed all-gather, UNIT bytes per thread and step
template <typename UNIT>
__global__ __launch_bounds__(256) void pack_rows_kernel(const PackArgs a, char* __restrict__ out) {
    const int64_t units_per_row = a.total / (int64_t)sizeof(UNIT);
    const int64_t n = a.rows * units_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / units_per_row, b = (i - r * units_per_row) * (int64_t)sizeof(UNIT);
        int p = 0;
#pragma unroll
        for (int q = 1; q < MAX_PARTS; ++q) p = (q < a.n && b >= a.off[q]) ? q : p;
        *(UNIT*)(out + r * a.total + b) = *(const UNIT*)(a.src[p] + r * a.row_bytes[p] + (b - a.off[p]));
    }
}

}  // namespace

extern "C" int mico_comm_unique_id(void* id_out) {
    MICO_CHECK(id_out != nullptr, "mico_comm_unique_id: null pointer");
    std::call_once(g_rccl_once, load_rccl);
    MICO_CHECK(g_rccl.handle != nullptr, "mico_comm_unique_id: %s", g_rccl_err ? g_rccl_err : "RCCL unavailable");
    nccl_uid id;
    RCCL_OK(g_rccl.GetUniqueId(&id), "mico_comm_unique_id");
    memcpy(id_out, &id, sizeof(id));
    return MICO_OK;
}

extern "C" int mico_comm_init(void** comm_out, int rank, int nranks, const void* id) {
    MICO_CHECK(comm_out && id && nranks >= 1 && rank >= 0 && rank < nranks, "mico_comm_init: bad arguments (rank %d of %d)", rank, nranks);
    std::call_once(g_rccl_once, load_rccl);
    MICO_CHECK(g_rccl.handle != nullptr, "mico_comm_init: %s", g_rccl_err ? g_rccl_err : "RCCL unavailable");
    nccl_uid uid;
    memcpy(&uid, id, sizeof(uid));
    Comm* c = new Comm{nullptr, rank, nranks};
    const int rc = g_rccl.CommInitRank(&c->c, nranks, uid, rank);     // (uses the calling thread's current HIP device)
    if (rc != 0) {
        delete c;
        return mico_set_err(MICO_ELAUNCH, "mico_comm_init: ncclCommInitRank failed with %d (%s)", rc, g_rccl.GetErrorString(rc));
    }
    *comm_out = c;
    return MICO_OK;
}

extern "C" int mico_comm_destroy(void* comm) {
    if (!comm) return MICO_OK;
    Comm* c = (Comm*)comm;
    const int rc = g_rccl.CommDestroy(c->c);
    delete c;
    MICO_CHECK(rc == 0, "mico_comm_destroy: ncclCommDestroy failed with %d", rc);
    return MICO_OK;
}

extern "C" int mico_comm_allgather(void* comm, const void* send, void* recv, int64_t bytes_per_rank, void* stream) {
    MICO_CHECK(comm && send && recv && bytes_per_rank >= 0, "mico_comm_allgather: bad arguments");
    if (bytes_per_rank == 0) return MICO_OK;
    Comm* c = (Comm*)comm;
    RCCL_OK(g_rccl.AllGather(send, recv, (size_t)bytes_per_rank, NCCL_UINT8, c->c, (hipStream_t)stream), "mico_comm_allgather");
    return MICO_OK;
}

extern "C" int mico_comm_allgather_packed(void* comm, const void* const* parts, const int64_t* row_bytes, int nparts, int64_t rows,
                                          void* pack_scratch, void* recv, void* stream) {
    MICO_CHECK(comm && parts && row_bytes && pack_scratch && recv && nparts >= 1 && nparts <= MAX_PARTS && rows >= 0,
               "mico_comm_allgather_packed: bad arguments (1 <= nparts <= %d)", MAX_PARTS);
    if (rows == 0) return MICO_OK;
    Comm* c = (Comm*)comm;
    PackArgs a{};
    a.n = nparts;
    a.rows = rows;
    int64_t off = 0;
    bool words = ((uintptr_t)pack_scratch & 3) == 0;
    for (int i = 0; i < nparts; ++i) {
        MICO_CHECK(parts[i] != nullptr && row_bytes[i] > 0, "mico_comm_allgather_packed: part %d is empty", i);
        a.src[i] = (const char*)parts[i];
        a.row_bytes[i] = row_bytes[i];
        a.off[i] = off;
        off += row_bytes[i];
        words = words && row_bytes[i] % 4 == 0 && ((uintptr_t)parts[i] & 3) == 0;
    }
    a.total = off;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = rows * (words ? off / 4 : off);
    const dim3 grid((unsigned)std::min<int64_t>((n + 255) / 256, 2048));
    if (words) MICO_LAUNCH((pack_rows_kernel<unsigned>), grid, dim3(256), 0, st, a, (char*)pack_scratch);
    else MICO_LAUNCH((pack_rows_kernel<unsigned char>), grid, dim3(256), 0, st, a, (char*)pack_scratch);
    MICO_LAUNCH_CHECK();
    RCCL_OK(g_rccl.AllGather(pack_scratch, recv, (size_t)(rows * off), NCCL_UINT8, c->c, st), "mico_comm_allgather_packed");
    return MICO_OK;
}

extern "C" int mico_comm_alltoallv(void* comm, const void* send, const int64_t* send_bytes, void* recv, const int64_t* recv_bytes, void* stream) {
    MICO_CHECK(comm && send_bytes && recv_bytes, "mico_comm_alltoallv: bad arguments");
    Comm* c = (Comm*)comm;
    hipStream_t st = (hipStream_t)stream;
    int64_t so = 0, ro = 0;
    // every argument is validated BEFORE the group opens, and an error inside it still closes it (a return between ncclGroupStart and
    // ncclGroupEnd left the group open and wedged the thread's later RCCL calls)
    for (int p = 0; p < c->nranks; ++p) {
        MICO_CHECK(send_bytes[p] >= 0 && recv_bytes[p] >= 0, "mico_comm_alltoallv: negative count for peer %d", p);
        MICO_CHECK(send_bytes[p] == 0 || send != nullptr, "mico_comm_alltoallv: null send buffer");
        MICO_CHECK(recv_bytes[p] == 0 || recv != nullptr, "mico_comm_alltoallv: null receive buffer");
    }
    RCCL_OK(g_rccl.GroupStart(), "mico_comm_alltoallv");
    int rc = 0;
    const char* where = "";
    for (int p = 0; p < c->nranks && rc == 0; ++p) {
        if (send_bytes[p] > 0) { rc = g_rccl.Send((const char*)send + so, (size_t)send_bytes[p], NCCL_UINT8, p, c->c, st); where = "send"; }
        if (rc == 0 && recv_bytes[p] > 0) { rc = g_rccl.Recv((char*)recv + ro, (size_t)recv_bytes[p], NCCL_UINT8, p, c->c, st); where = "recv"; }
        so += send_bytes[p];
        ro += recv_bytes[p];
    }
    const int rc_end = g_rccl.GroupEnd();
    if (rc != 0) return mico_set_err(MICO_ELAUNCH, "mico_comm_alltoallv (%s): RCCL error %d (%s)", where, rc, g_rccl.GetErrorString(rc));
    RCCL_OK(rc_end, "mico_comm_alltoallv");
    return MICO_OK;
}

extern "C" int mico_comm_allreduce_f32(void* comm, float* buf, int64_t count, int average, void* stream) {
    MICO_CHECK(comm && (buf || count == 0) && count >= 0, "mico_comm_allreduce_f32: bad arguments");
    if (count == 0) return MICO_OK;
    Comm* c = (Comm*)comm;
    RCCL_OK(g_rccl.AllReduce(buf, buf, (size_t)count, NCCL_FLOAT32, average ? NCCL_AVG : NCCL_SUM, c->c, (hipStream_t)stream), "mico_comm_allreduce_f32");
    return MICO_OK;
}

extern "C" int mico_comm_reduce_scatter_f32(void* comm, const float* send, float* recv, int64_t count_per_rank, int average, void* stream) {
    MICO_CHECK(comm && send && recv && count_per_rank >= 0, "mico_comm_reduce_scatter_f32: bad arguments");
    if (count_per_rank == 0) return MICO_OK;
    Comm* c = (Comm*)comm;
    RCCL_OK(g_rccl.ReduceScatter(send, recv, (size_t)count_per_rank, NCCL_FLOAT32, average ? NCCL_AVG : NCCL_SUM, c->c, (hipStream_t)stream),
            "mico_comm_reduce_scatter_f32");
    return MICO_OK;
}
