// alga_amd/csrc/ingest_kernels.hip -- duplicate / prefix-read removal and id compaction on the GPU.
//
// Reference (paths relative to the reference root):
//   ReadPreprocess::getSortedReads        src/IO/ReadPreprocess.cpp:115-132   order: bit string with bit 0 most significant, then
//                                                                               length, then id
//   ReadPreprocess::removePrefixReads...  src/IO/ReadPreprocess.cpp:13-77     adjacent pairs (a, b) of that order: a is removed when
//                                                                               it is a prefix of b (mode 2; its twin too when shorter)
//                                                                               or equal to b (mode 1)
//   id compaction, pairedReadOffset       src/main.cpp:150-232
//   removal of too-short reads            src/main.cpp:253-266
// Host statement of the same stage: alga_amd/host/ingest.cpp preprocess_host (checked against the oracle on the CPU).
//
// The order is a lexicographic one on zero-padded rows, so it is produced by a least-significant-digit radix sort: one stable
// 64-bit pass per pair of bit-reversed words from the last pair to the first, after a pass on the length; equal keys keep the
// id order of the start permutation.  Nodes the parser removed carry all-ones keys in every pass and end up behind the live ones.
#include <hip/hip_runtime.h>
#include <algorithm>

#include "prefsuf_common.h"
#include "ingest_kernels.h"

namespace alga {

static inline unsigned grid_for(uint64_t n, int block, unsigned cap) {
    return (unsigned) std::max<uint64_t>(1, std::min<uint64_t>((n + (uint64_t) block - 1) / (uint64_t) block, cap));
}

__global__ void __launch_bounds__(256) k_pp_iota(uint32_t *__restrict__ perm, uint64_t n) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) perm[i] = (uint32_t) i;
}

// pass < 0: key = length; pass k >= 0: key = bit-reversed words 2k, 2k+1 (bit 0 of the row is the most significant bit of the order)
__global__ void __launch_bounds__(256) k_pp_keys(const uint32_t *__restrict__ rows, int stride, int used_words, const int32_t *__restrict__ len,
                                                  const uint32_t *__restrict__ perm, uint64_t n, int pass, unsigned long long *__restrict__ keys) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t id = perm[i];
        const int l = len[id];
        unsigned long long k = ~0ull;
        if (l >= 0) {
            if (pass < 0) k = (unsigned long long) (uint32_t) l;
            else {
                const uint32_t *r = rows + (size_t) id * stride;
                const uint32_t a = __brev(r[2 * pass]);
                const uint32_t b = 2 * pass + 1 < used_words ? __brev(r[2 * pass + 1]) : 0u;
                k = ((unsigned long long) a << 32) | b;
            }
        }
        keys[i] = k;
    }
}

// adjacent pairs of the sorted live nodes (ReadPreprocess.cpp:30-62)
__global__ void __launch_bounds__(256) k_pp_mark(const uint32_t *__restrict__ rows, int stride, const int32_t *__restrict__ len,
                                                  const uint32_t *__restrict__ perm, uint64_t n_live, int mode, uint8_t *__restrict__ mark) {
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i + 1 < n_live; i += (uint64_t) gridDim.x * blockDim.x) {
        const uint32_t a = perm[i], b = perm[i + 1];
        const int la = len[a], lb = len[b];
        const uint32_t *wa = rows + (size_t) a * stride, *wb = rows + (size_t) b * stride;
        const int m = min(blocks_of(la), blocks_of(lb));
        long long ind = 1000000000ll;                                   // Bitset::mismatch (Bitset.cpp:858-877)
        for (int q = 0; q < m; q++) {
            const uint32_t x = wa[q] ^ wb[q];
            if (x) { ind = (long long) q * 32 + (__ffs((int) x) - 1); break; }
        }
        const long long ms = 2ll * (long long) min(la, lb);
        const int l = (int) ((ind < ms ? ind : ms) >> 1);
        if (mode == 1) { if (l == la && la == lb) mark[a] = 1; }
        else if (l == la) {
            mark[a] = 1;
            if (la < lb) mark[a ^ 1u] = 1;                               // a proper prefix takes its twin with it
        }
    }
}

// marked nodes become removed; keep[r] = read r survives; tallies: [0] nodes removed here, [1] twin inconsistencies, [2] max length kept
__global__ void __launch_bounds__(256) k_pp_apply(int32_t *__restrict__ len, const uint8_t *__restrict__ mark, uint64_t n_reads,
                                                   uint32_t *__restrict__ keep, unsigned long long *__restrict__ tally) {
    unsigned long long removed = 0, bad = 0, mx = 0;
    for (uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; r < n_reads; r += (uint64_t) gridDim.x * blockDim.x) {
        int l0 = len[2 * r], l1 = len[2 * r + 1];
        if (mark && l0 >= 0 && mark[2 * r]) { l0 = -1; removed++; }
        if (mark && l1 >= 0 && mark[2 * r + 1]) { l1 = -1; removed++; }
        len[2 * r] = l0; len[2 * r + 1] = l1;
        // The reference's compaction looks at the EVEN node of a pair only (src/main.cpp:165-171): present -> the odd one must be
        // present too (it asserts); absent -> the pair is dropped, whatever the odd node is.  The second case is not an input
        // error: a read that equals its own reverse complement loses its even node as the "duplicate" of the odd one
        // (src/IO/ReadPreprocess.cpp:13-77) and the reference then drops the read.  Both are reproduced as they are.
        keep[r] = l0 >= 0 ? 1u : 0u;
        if (l0 >= 0 && l1 < 0) bad++;
        if (l0 >= 0 && (unsigned long long) l0 > mx) mx = (unsigned long long) l0;
        if (l0 >= 0 && l1 >= 0 && (unsigned long long) l1 > mx) mx = (unsigned long long) l1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        removed += __shfl_xor(removed, o);
        bad += __shfl_xor(bad, o);
        const unsigned long long t = __shfl_xor(mx, o);
        mx = t > mx ? t : mx;
    }
    __shared__ unsigned long long s_r[4], s_b[4], s_m[4];
    const int wv = (int) (threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0) { s_r[wv] = removed; s_b[wv] = bad; s_m[wv] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) { removed += s_r[k]; bad += s_b[k]; mx = s_m[k] > mx ? s_m[k] : mx; }
        if (removed) atomicAdd(&tally[0], removed);
        if (bad) atomicAdd(&tally[1], bad);
        if (mx) atomicMax(&tally[2], mx);
    }
}

// surviving reads move to their compacted ids (src/main.cpp:150-232): one wave-quarter per node row would be overkill, rows are
// short: a thread copies one node; pairedReadOffset: reads 2q and 2q+1 of the input are "mates" when both survive.
__global__ void __launch_bounds__(256) k_pp_compact(const uint32_t *__restrict__ rows, int stride_in, const int32_t *__restrict__ len,
                                                     const uint32_t *__restrict__ keep, const uint32_t *__restrict__ pos, uint64_t n_reads,
                                                     int min_keep_len, uint32_t *__restrict__ out_rows, int stride_out, int32_t *__restrict__ out_len,
                                                     uint8_t *__restrict__ out_pair, unsigned long long *__restrict__ tally) {
    unsigned long long shorted = 0;
    for (uint64_t t = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; t < 2 * n_reads; t += (uint64_t) gridDim.x * blockDim.x) {
        const uint64_t r = t >> 1;
        if (!keep[r]) continue;
        const uint64_t dst = 2ull * pos[r] + (t & 1ull);
        int l = len[t];
        const uint32_t *src = rows + (size_t) t * stride_in;
        uint32_t *o = out_rows + (size_t) dst * stride_out;
        const bool drop = l < min_keep_len;                               // src/main.cpp:253-266: the node stays, emptied
        const int nb = drop ? 0 : blocks_of(l);
        for (int q = 0; q < stride_out; q++) o[q] = q < nb ? src[q] : 0u;
        out_len[dst] = drop ? 0 : l;
        uint8_t po = 0;
        if ((r & 1ull) == 0) po = (r + 1 < n_reads && keep[r + 1]) ? 1 : 0;
        else po = keep[r - 1] ? 2 : 0;
        out_pair[dst] = po;
        shorted += drop;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) shorted += __shfl_xor(shorted, o);
    __shared__ unsigned long long s_s[4];
    const int wv = (int) (threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0) s_s[wv] = shorted;
    __syncthreads();
    if (threadIdx.x == 0) { shorted = s_s[0] + s_s[1] + s_s[2] + s_s[3]; if (shorted) atomicAdd(&tally[3], shorted); }
}

void launch_pp_iota(uint32_t *perm, uint64_t n, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pp_iota, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, perm, n);
}
void launch_pp_keys(const uint32_t *rows, int stride, int used_words, const int32_t *len, const uint32_t *perm, uint64_t n, int pass,
                    unsigned long long *keys, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_pp_keys, dim3(grid_for(n, 256, 8192)), dim3(256), 0, s, rows, stride, used_words, len, perm, n, pass, keys);
}
void launch_pp_mark(const uint32_t *rows, int stride, const int32_t *len, const uint32_t *perm, uint64_t n_live, int mode, uint8_t *mark, hipStream_t s) {
    if (n_live > 1) hipLaunchKernelGGL(k_pp_mark, dim3(grid_for(n_live, 256, 8192)), dim3(256), 0, s, rows, stride, len, perm, n_live, mode, mark);
}
void launch_pp_apply(int32_t *len, const uint8_t *mark, uint64_t n_reads, uint32_t *keep, unsigned long long *tally, hipStream_t s) {
    if (n_reads) hipLaunchKernelGGL(k_pp_apply, dim3(grid_for(n_reads, 256 * 4, 1024)), dim3(256), 0, s, len, mark, n_reads, keep, tally);
}
void launch_pp_compact(const uint32_t *rows, int stride_in, const int32_t *len, const uint32_t *keep, const uint32_t *pos, uint64_t n_reads,
                       int min_keep_len, uint32_t *out_rows, int stride_out, int32_t *out_len, uint8_t *out_pair, unsigned long long *tally, hipStream_t s) {
    if (n_reads) hipLaunchKernelGGL(k_pp_compact, dim3(grid_for(2 * n_reads, 256 * 4, 1024)), dim3(256), 0, s, rows, stride_in, len, keep, pos, n_reads,
                                    min_keep_len, out_rows, stride_out, out_len, out_pair, tally);
}

} // namespace alga
