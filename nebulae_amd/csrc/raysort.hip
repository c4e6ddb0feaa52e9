// raysort.hip -- stable LSD radix sort of (key, value) pairs for the GI ray reordering (gi.hip).
//
// The keys are <= 16 bits (Morton cells of ray origins) and there are ~2 M of them per frame, which is small enough
// that a general-purpose device sort spends most of its time outside its sorting passes (hipcub/rocprim onesweep on
// this input: 2 x 29 us of passes + 32 us histogram + five 5-us memsets = ~100 us).  This one is cut to the job:
// 8-bit digits, and per pass three kernels on the launch stream with no memsets and no look-back spinning:
//
//   rs_hist_kernel    : per tile (4096 pairs) digit histogram                       -> hist[digit][tile]
//   rs_scan_kernel    : one workgroup per digit, exclusive scan across the tiles    -> hist (in place), total[digit]
//   rs_scatter_kernel : re-reads the tile, ranks every pair among the equal digits before it (ballot match inside a
//                       wave, LDS tables across waves), sorts the tile by digit in LDS and writes each digit's run to
//                       its global position -- runs of ~16 pairs, so the scattered writes still fill 64-byte sectors.
//
// Stability (pass 2 must keep pass 1's order) comes from the ranking order: wave w of a tile owns pairs
// [256 w, 256 (w+1)), walks them in 4 rounds of 64, and lanes rank in lane order.
#include "neb_internal.h"

namespace neb {

namespace {

constexpr int kRsThreads = 1024;               // 16 waves per tile: the ranking is serial inside a wave, so many short waves
constexpr int kRsItems = 4;
constexpr int kRsTile = kRsThreads * kRsItems; // 4096 pairs per workgroup
constexpr int kRsWaves = kRsThreads / 64;
constexpr int kRsRounds = kRsTile / kRsWaves / 64; // 4 rounds of 64 pairs per wave

// lanes of the wave whose 8-bit digit equals this lane's (invalid lanes match nobody)
__device__ __forceinline__ unsigned long long rs_match(uint32_t digit, bool valid)
{
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (digit >> b) & 1u;
        const unsigned long long m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    return peers;
}

__device__ __forceinline__ uint32_t rs_rank_below(unsigned long long peers) // set bits of peers below this lane
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
}

__global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t mask,
                                                             uint32_t* __restrict__ hist, uint32_t n_tiles)
{
    __shared__ uint32_t h[256];
    const uint32_t tid = threadIdx.x;
    if (tid < 256)
        h[tid] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kRsTile;
#pragma unroll
    for (int k = 0; k < kRsItems; ++k) {
        const uint32_t idx = base + k * kRsThreads + tid;
        if (idx < n)
            atomicAdd(&h[(keys[idx] >> shift) & mask], 1u); // LDS atomic, no return
    }
    __syncthreads();
    if (tid < 256)
        hist[(size_t)tid * n_tiles + blockIdx.x] = h[tid];
}

// exclusive scan of 512 values held one per thread
__device__ __forceinline__ uint32_t rs_block_exclusive_scan_512(uint32_t v, uint32_t* wave_tot /*[8]*/, uint32_t& total)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= (uint32_t)off)
            inc += t;
    }
    if (lane == 63)
        wave_tot[wave] = inc;
    __syncthreads();
    uint32_t prefix = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const uint32_t t = wave_tot[w];
        if ((uint32_t)w < wave)
            prefix += t;
        sum += t;
    }
    __syncthreads(); // wave_tot is reused by the caller's next chunk
    total = sum;
    return prefix + inc - v;
}

__global__ __launch_bounds__(512) void rs_scan_kernel(uint32_t* __restrict__ hist, uint32_t n_tiles, uint32_t* __restrict__ digit_total)
{
    __shared__ uint32_t wave_tot[8];
    uint32_t* row = hist + (size_t)blockIdx.x * n_tiles;
    uint32_t carry = 0;
    for (uint32_t c = 0; c < n_tiles; c += 512) {
        const uint32_t i = c + threadIdx.x;
        const uint32_t v = i < n_tiles ? row[i] : 0u;
        uint32_t total;
        const uint32_t ex = rs_block_exclusive_scan_512(v, wave_tot, total);
        if (i < n_tiles)
            row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0)
        digit_total[blockIdx.x] = carry;
}

__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                uint32_t n, uint32_t shift, uint32_t mask, const uint32_t* __restrict__ hist,
                                                                const uint32_t* __restrict__ digit_total, uint32_t n_tiles,
                                                                uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out)
{
    __shared__ uint32_t s_keys[kRsTile];
    __shared__ uint32_t s_vals[kRsTile];
    __shared__ uint32_t woff[kRsWaves][256]; // per wave: count of a digit, then the running local position of its next pair
    __shared__ uint32_t gdelta[256];         // global position of a digit's first pair of this tile minus its local position
    __shared__ uint32_t scan_tmp[4];
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t tile0 = blockIdx.x * kRsTile;
    const uint32_t base = tile0 + wave * (kRsTile / kRsWaves); // wave w owns pairs [256 w, 256 (w + 1)) of the tile

    uint32_t key[kRsRounds], val[kRsRounds];
#pragma unroll
    for (int r = 0; r < kRsRounds; ++r) {
        const uint32_t idx = base + r * 64 + lane;
        key[r] = idx < n ? keys_in[idx] : 0xffffffffu;
        val[r] = idx < n ? vals_in[idx] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 256 / 64; ++k)
        woff[wave][k * 64 + lane] = 0;
    // (only this wave touches woff[wave] until the barrier, and a wave's LDS operations execute in order)
    unsigned long long peers[kRsRounds];
#pragma unroll
    for (int r = 0; r < kRsRounds; ++r) {
        const bool valid = base + r * 64 + lane < n;
        const uint32_t d = (key[r] >> shift) & mask;
        peers[r] = rs_match(d, valid);
        if (valid && rs_rank_below(peers[r]) == 0)
            woff[wave][d] += (uint32_t)__popcll(peers[r]);
    }
    __syncthreads();
    // thread t < 256 owns digit t: local start of the digit in the sorted tile (exclusive scan over the digits of the
    // tile's counts), the per-wave starts inside it, and the offset to the digit's global run
    uint32_t cnt = 0, dt = 0, inc = 0, dinc = 0;
    if (tid < 256) {
#pragma unroll
        for (int w = 0; w < kRsWaves; ++w)
            cnt += woff[w][tid];
        dt = digit_total[tid];
        inc = cnt;
        dinc = dt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { // waves 0..3 are complete, so the shuffles see all their lanes
            const uint32_t t0 = __shfl_up(inc, off), t1 = __shfl_up(dinc, off);
            if (lane >= (uint32_t)off) {
                inc += t0;
                dinc += t1;
            }
        }
        if (lane == 63)
            scan_tmp[wave] = inc;
    }
    __syncthreads();
    uint32_t lstart = 0;
    if (tid < 256) {
        lstart = inc - cnt;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if ((uint32_t)w < wave)
                lstart += scan_tmp[w];
    }
    __syncthreads();
    if (tid < 256 && lane == 63)
        scan_tmp[wave] = dinc;
    __syncthreads();
    if (tid < 256) {
        // global base of digit t: pairs with smaller digits anywhere + pairs with this digit in earlier tiles
        uint32_t gbase = dinc - dt;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if ((uint32_t)w < wave)
                gbase += scan_tmp[w];
        gbase += hist[(size_t)tid * n_tiles + blockIdx.x];
        gdelta[tid] = gbase - lstart;
        uint32_t run = lstart;
#pragma unroll
        for (int w = 0; w < kRsWaves; ++w) {
            const uint32_t c = woff[w][tid];
            woff[w][tid] = run;
            run += c;
        }
    }
    __syncthreads();
    // rank every pair and drop it at its place in the tile's digit order
#pragma unroll
    for (int r = 0; r < kRsRounds; ++r) {
        const bool valid = base + r * 64 + lane < n;
        const uint32_t d = (key[r] >> shift) & mask;
        const uint32_t rank = rs_rank_below(peers[r]);
        if (valid) {
            const uint32_t pos = woff[wave][d] + rank;
            s_keys[pos] = key[r];
            s_vals[pos] = val[r];
        }
        // (the leaders move the running positions on only after every lane has read its digit's: same instruction
        // stream, LDS operations of a wave execute in order)
        if (valid && rank == 0)
            woff[wave][d] += (uint32_t)__popcll(peers[r]);
    }
    __syncthreads();
    const uint32_t n_here = min((uint32_t)kRsTile, n - tile0);
#pragma unroll
    for (int k = 0; k < kRsItems; ++k) {
        const uint32_t j = k * kRsThreads + tid;
        if (j < n_here) {
            const uint32_t kk = s_keys[j];
            const uint32_t dst = gdelta[(kk >> shift) & mask] + j;
            if (keys_out)
                keys_out[dst] = kk;
            vals_out[dst] = s_vals[j];
        }
    }
}

} // namespace

size_t ray_sort_scratch_bytes(size_t n)
{
    const size_t n_tiles = (n + kRsTile - 1) / kRsTile;
    return (256 * n_tiles + 256) * sizeof(uint32_t);
}

// Sorts (keys, vals)[0, n) by key bits [0, bits), bits <= 16, stable.  keys / vals are overwritten (they serve as the
// ping-pong partner of keys_tmp / vals_tmp); the sorted values land in vals_out, which may alias vals.
hipError_t ray_sort_pairs(uint32_t* keys, uint32_t* vals, uint32_t* keys_tmp, uint32_t* vals_tmp, uint32_t* vals_out, size_t n, int bits,
                          void* scratch, hipStream_t stream)
{
    if (n == 0)
        return hipSuccess;
    const uint32_t n_tiles = (uint32_t)((n + kRsTile - 1) / kRsTile);
    uint32_t* hist = static_cast<uint32_t*>(scratch);
    uint32_t* digit_total = hist + (size_t)256 * n_tiles;
    const int passes = bits <= 8 ? 1 : 2;
    const uint32_t* kin = keys;
    const uint32_t* vin = vals;
    for (int p = 0; p < passes; ++p) {
        const bool last = p + 1 == passes;
        const uint32_t mask = (1u << (bits - 8 * p < 8 ? bits - 8 * p : 8)) - 1u; // key bits at and above `bits` are ignored
        uint32_t* kout = last ? nullptr : keys_tmp;
        uint32_t* vout = last ? vals_out : vals_tmp;
        if (last && vout == vin) // a single pass cannot sort in place
            vout = vals_tmp;
        hipLaunchKernelGGL(rs_hist_kernel, dim3(n_tiles), dim3(kRsThreads), 0, stream, kin, (uint32_t)n, (uint32_t)(8 * p), mask, hist, n_tiles);
        hipLaunchKernelGGL(rs_scan_kernel, dim3(256), dim3(512), 0, stream, hist, n_tiles, digit_total);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(n_tiles), dim3(kRsThreads), 0, stream, kin, vin, (uint32_t)n, (uint32_t)(8 * p), mask, hist,
                           digit_total, n_tiles, kout, vout);
        if (last && vout != vals_out) {
            hipError_t e = hipMemcpyAsync(vals_out, vout, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess)
                return e;
        }
        kin = keys_tmp;
        vin = vals_tmp;
    }
    return hipGetLastError();
}

} // namespace neb

extern "C" int neb_debug_sort_pairs(neb_ctx* ctx, const uint32_t* keys, const uint32_t* values, uint32_t n, int bits, uint32_t* sorted_values)
{
    if (!ctx)
        return NEB_ERR_INVALID_ARG;
    if (!keys || !values || !sorted_values || bits < 1 || bits > 16) {
        ctx->last_error = "neb_debug_sort_pairs: null array or bits outside 1..16";
        return NEB_ERR_INVALID_ARG;
    }
    if (n == 0)
        return NEB_OK;
    neb::DeviceGuard guard(ctx->device);
    hipError_t e = guard.err;
    uint32_t* d = nullptr; // {keys, vals, keys_tmp, vals_tmp}
    void* scratch = nullptr;
    const size_t bytes = (size_t)n * sizeof(uint32_t);
    if (e == hipSuccess)
        e = hipMalloc(reinterpret_cast<void**>(&d), 4 * bytes);
    if (e == hipSuccess)
        e = hipMalloc(&scratch, neb::ray_sort_scratch_bytes(n));
    if (e == hipSuccess)
        e = hipMemcpy(d, keys, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d + n, values, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = neb::ray_sort_pairs(d, d + n, d + 2 * (size_t)n, d + 3 * (size_t)n, d + n, n, bits, scratch, nullptr);
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e == hipSuccess)
        e = hipMemcpy(sorted_values, d + n, bytes, hipMemcpyDeviceToHost);
    if (d)
        (void)hipFree(d);
    if (scratch)
        (void)hipFree(scratch);
    if (e != hipSuccess) {
        ctx->last_error = std::string("neb_debug_sort_pairs: ") + hipGetErrorString(e);
        return NEB_ERR_HIP;
    }
    return NEB_OK;
}
