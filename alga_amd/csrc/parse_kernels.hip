// alga_amd/csrc/parse_kernels.hip -- input stage N2 on the GPU: file bytes -> packed node rows.
//
// What InputReader does per record (src/IO/InputReader.cpp:142-180 readOneRead1, :272-391 readParallelJob; reference paths
// relative to its root), for FASTA (two lines per record) and FASTQ (four): the sequence line, leading blanks dropped and cut at
// the first blank (:288-293), READ_END_TRIM_LEFT/RIGHT nucleotides off each end unless the read is shorter than their sum + 10
// (:298-303), letters checked (:320-336), reads with N dropped (REMOVE_READS_WITH_N = 1), reads whose minimal period is <= 20
// dropped (:341-354, include/Utils/MyUtils.h:160-170), the read 2-bit packed (src/DataStructures/Read.cpp:40-68) together with
// its reverse complement (:363-377).  Node order: record k -> nodes 2k (reverse complement), 2k + 1 (forward) (:78-80); with two
// files record i of file f is read 2i + f (:53-76).  Input ends at the first empty sequence line (:284).
//   k_nl_count / k_nl_write   positions of the line ends (std::getline semantics: '\n' only)
//   k_line_stats              longest sequence line, first record with an empty sequence line
//   k_parse_records           one thread per record: everything above, rows and lengths out
#include <hip/hip_runtime.h>
#include <algorithm>
#include "parse_kernels.h"

namespace alga {

constexpr int NL_THREADS = 256;
constexpr int NL_BYTES = 16;                               // bytes per thread
constexpr int NL_TILE = NL_THREADS * NL_BYTES;

__device__ __forceinline__ uint32_t nl_mask(uint32_t w) {  // 0x80 in every byte of w that is '\n' (exact, no borrow effects)
    const uint32_t x = w ^ 0x0A0A0A0Au;
    const uint32_t t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
    return ~(t | x | 0x7F7F7F7Fu);
}

__device__ __forceinline__ void load16(const uint8_t *__restrict__ b, uint64_t n, uint64_t off, uint32_t (&w)[4]) {
    if (off + 16 <= n) { const uint4 v = *reinterpret_cast<const uint4 *>(b + off); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; return; }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { const uint64_t p = off + 4 * k + j; if (p < n) x |= (uint32_t) b[p] << (8 * j); }
        w[k] = x;
    }
}

__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t *total, uint32_t *lds /* >= 8 words */) {
    const int lane = (int) (threadIdx.x & 63u), wave = (int) (threadIdx.x >> 6);
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t) __shfl_up((int) inc, o); if (lane >= o) inc += t; }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t off = 0, tot = 0;
    for (int w = 0; w < (int) (blockDim.x >> 6); w++) { const uint32_t s = lds[w]; if (w < wave) off += s; tot += s; }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}

__global__ void __launch_bounds__(NL_THREADS) k_nl_count(const uint8_t *__restrict__ b, uint64_t n, uint32_t *__restrict__ tile_cnt) {
    __shared__ uint32_t lds[8];
    const uint64_t off = (uint64_t) blockIdx.x * NL_TILE + (uint64_t) threadIdx.x * NL_BYTES;
    uint32_t c = 0;
    if (off < n) { uint32_t w[4]; load16(b, n, off, w); c = __popc(nl_mask(w[0])) + __popc(nl_mask(w[1])) + __popc(nl_mask(w[2])) + __popc(nl_mask(w[3])); }
    uint32_t tot;
    block_scan_excl(c, &tot, lds);
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(NL_THREADS) k_nl_write(const uint8_t *__restrict__ b, uint64_t n, const uint32_t *__restrict__ tile_off,
                                                          unsigned long long *__restrict__ nl_pos) {
    __shared__ uint32_t lds[8];
    const uint64_t off = (uint64_t) blockIdx.x * NL_TILE + (uint64_t) threadIdx.x * NL_BYTES;
    uint32_t w[4] = {0, 0, 0, 0}, c = 0;
    if (off < n) { load16(b, n, off, w); c = __popc(nl_mask(w[0])) + __popc(nl_mask(w[1])) + __popc(nl_mask(w[2])) + __popc(nl_mask(w[3])); }
    uint32_t tot;
    uint64_t dst = (uint64_t) tile_off[blockIdx.x] + block_scan_excl(c, &tot, lds);
    if (c) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t m = nl_mask(w[k]);
            while (m) { const int bit = __ffs((int) m) - 1; m &= m - 1; nl_pos[dst++] = off + 4 * k + (uint64_t) (bit >> 3); }
        }
    }
}

struct LineIndex {
    const unsigned long long *nl_pos;
    uint64_t n_nl, n_bytes, n_lines;
    __device__ __forceinline__ void span(uint64_t line, uint64_t &beg, uint64_t &end) const {
        beg = line == 0 ? 0 : nl_pos[line - 1] + 1;
        end = line < n_nl ? nl_pos[line] : n_bytes;
    }
};

// out[0] = longest sequence line, out[1] = first record whose sequence line is empty
__global__ void __launch_bounds__(256) k_line_stats(LineIndex li, int lines_per_record, uint64_t n_cand, unsigned long long *__restrict__ out) {
    unsigned long long mx = 0, first = ~0ull;
    for (uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; r < n_cand; r += (uint64_t) gridDim.x * blockDim.x) {
        uint64_t b, e;
        li.span(r * (uint64_t) lines_per_record + 1, b, e);
        const unsigned long long l = e - b;
        mx = l > mx ? l : mx;
        if (l == 0 && r < first) first = r;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long a = __shfl_xor(mx, o), c = __shfl_xor(first, o);
        mx = a > mx ? a : mx; first = c < first ? c : first;
    }
    if ((threadIdx.x & 63u) == 0) { if (mx) atomicMax(&out[0], mx); if (first != ~0ull) atomicMin(&out[1], first); }
}

// character classes of a sequence line: 0..3 = A C G T (the 2-bit codes), 4 = N, 5 = U, 255 = anything else
__device__ __forceinline__ uint32_t char_class(uint8_t c) {
    return c == 'A' ? 0u : (c == 'C' ? 1u : (c == 'G' ? 2u : (c == 'T' ? 3u : (c == 'N' ? 4u : (c == 'U' ? 5u : 255u)))));
}

// tally: [0] records with N, [1] STR records, [2] sum of the kept lengths (per read), [3] kept reads, [4] first bad record << 8 | its character
__global__ void __launch_bounds__(256) k_parse_records(const uint8_t *__restrict__ bytes, LineIndex li, int lines_per_record, uint64_t n_rec,
                                                        int file_index, int paired, ParseCfg pc, uint32_t *__restrict__ rows, int W,
                                                        int32_t *__restrict__ len, unsigned long long *__restrict__ tally) {
    const uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long t_n = 0, t_str = 0, t_sum = 0, t_kept = 0;
    if (r < n_rec) {
        uint64_t b, e;
        li.span(r * (uint64_t) lines_per_record + 1, b, e);
        while (b < e && bytes[b] == ' ') b++;              // :288-293
        uint64_t q = b;
        while (q < e && bytes[q] != ' ') q++;
        int n = (int) (q - b);
        if (!(n < pc.trim_left + pc.trim_right + 10)) {    // :298-303
            const int l = min(pc.trim_left, n);
            b += (uint64_t) l; n -= l;
            n -= min(pc.trim_right, n);
        }
        const uint8_t *s = bytes + b;
        const uint64_t k = paired ? 2 * r + (uint64_t) file_index : r;       // read slot in node order
        uint32_t *fw = rows + (2 * k + 1) * (size_t) W, *rc = rows + (2 * k) * (size_t) W;
        bool hasN = false, bad = false;
        uint8_t badc = 0;
        for (int i = 0; i < n; i++) {                      // :320-336
            const uint32_t c = char_class(s[i]);
            if (c == 255u) { if (!bad) { bad = true; badc = s[i]; } }
            hasN |= c == 4u;
        }
        // the code a letter packs as and compares as: U stays apart from T unless --rna (then it IS T); anything else than A C G T packs as A
        auto code = [&](int i) -> uint32_t { const uint32_t c = char_class(s[i]); return c == 5u ? (pc.rna ? 3u : 6u) : c; };
        int L = -1;
        if (bad) atomicMin(&tally[4], (unsigned long long) r << 8 | badc);
        else if (hasN) t_n = 1;
        else {
            bool str = n <= 20;                            // a string is periodic with its own length: min period <= n
            for (int p = 1; p <= 20 && !str; p++) {        // any period <= 20 bounds the minimal one
                int i = 0;
                while (i < n - p && code(i) == code(i + p)) i++;
                str = i == n - p;
            }
            if (str) t_str = 1;
            else {
                L = n; t_sum = (unsigned long long) n; t_kept = 1;
                for (int wq = 0; wq < W; wq++) {
                    uint32_t vf = 0, vr = 0;
                    for (int j = 0; j < 16; j++) {
                        const int i = 16 * wq + j;
                        if (i < n) {
                            const uint32_t cf = code(i), cr = code(n - 1 - i);
                            vf |= (cf < 4u ? cf : 0u) << (2 * j);
                            vr |= (cr < 4u ? 3u - cr : 0u) << (2 * j);      // getComplimentaryString, :23-33
                        }
                    }
                    fw[wq] = vf; rc[wq] = vr;
                }
            }
        }
        if (L < 0) for (int wq = 0; wq < W; wq++) { fw[wq] = 0u; rc[wq] = 0u; }
        len[2 * k] = L; len[2 * k + 1] = L;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { t_n += __shfl_xor(t_n, o); t_str += __shfl_xor(t_str, o); t_sum += __shfl_xor(t_sum, o); t_kept += __shfl_xor(t_kept, o); }
    if ((threadIdx.x & 63u) == 0) {
        if (t_n) atomicAdd(&tally[0], t_n);
        if (t_str) atomicAdd(&tally[1], t_str);
        if (t_sum) atomicAdd(&tally[2], t_sum);
        if (t_kept) atomicAdd(&tally[3], t_kept);
    }
}

// max length / live count of a device length array (the N1 stage sizes its passes from them)
__global__ void __launch_bounds__(256) k_len_stats(const int32_t *__restrict__ len, uint64_t n, unsigned long long *__restrict__ out /* [0] max, [1] live */) {
    unsigned long long mx = 0, live = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
        const int l = len[i];
        if (l >= 0) { live++; mx = (unsigned long long) l > mx ? (unsigned long long) l : mx; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long a = __shfl_xor(mx, o); mx = a > mx ? a : mx; live += __shfl_xor(live, o); }
    if ((threadIdx.x & 63u) == 0) { if (mx) atomicMax(&out[0], mx); if (live) atomicAdd(&out[1], live); }
}

uint64_t nl_tiles(uint64_t n_bytes) { return (n_bytes + NL_TILE - 1) / NL_TILE; }
void launch_nl_count(const uint8_t *b, uint64_t n, uint32_t *tile_cnt, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_nl_count, dim3((unsigned) nl_tiles(n)), dim3(NL_THREADS), 0, s, b, n, tile_cnt);
}
void launch_nl_write(const uint8_t *b, uint64_t n, const uint32_t *tile_off, unsigned long long *nl_pos, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_nl_write, dim3((unsigned) nl_tiles(n)), dim3(NL_THREADS), 0, s, b, n, tile_off, nl_pos);
}
void launch_line_stats(const unsigned long long *nl_pos, uint64_t n_nl, uint64_t n_bytes, uint64_t n_lines, int lines_per_record, uint64_t n_cand,
                       unsigned long long *out, hipStream_t s) {
    if (!n_cand) return;
    LineIndex li{nl_pos, n_nl, n_bytes, n_lines};
    hipLaunchKernelGGL(k_line_stats, dim3((unsigned) std::min<uint64_t>((n_cand + 255) / 256, 4096)), dim3(256), 0, s, li, lines_per_record, n_cand, out);
}
void launch_parse_records(const uint8_t *bytes, const unsigned long long *nl_pos, uint64_t n_nl, uint64_t n_bytes, uint64_t n_lines, int lines_per_record,
                          uint64_t n_rec, int file_index, int paired, const ParseCfg &pc, uint32_t *rows, int W, int32_t *len, unsigned long long *tally,
                          hipStream_t s) {
    if (!n_rec) return;
    LineIndex li{nl_pos, n_nl, n_bytes, n_lines};
    hipLaunchKernelGGL(k_parse_records, dim3((unsigned) ((n_rec + 255) / 256)), dim3(256), 0, s, bytes, li, lines_per_record, n_rec, file_index, paired, pc,
                       rows, W, len, tally);
}
void launch_len_stats(const int32_t *len, uint64_t n, unsigned long long *out, hipStream_t s) {
    if (n) hipLaunchKernelGGL(k_len_stats, dim3((unsigned) std::min<uint64_t>((n + 255) / 256, 2048)), dim3(256), 0, s, len, n, out);
}

} // namespace alga
