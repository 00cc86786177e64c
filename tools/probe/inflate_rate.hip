// What would a device inflate give f3?  The decode loop every table-driven inflate spends its time in -- refill a bit
// buffer, look the next bits up in a Huffman table, emit a literal -- run by ONE lane of a wave per gzip stream (a DEFLATE stream is
// sequential: the position of symbol i+1 is known once symbol i is decoded), with everything around it done the way a real device
// decoder would: the wave fetches the compressed stream in coalesced 512-byte chunks one chunk ahead, the table and both rings live in
// LDS, the output leaves in coalesced words.  The text is FASTH-shaped (header, bases, qualities, barcode lines of the synthetic files),
// the code is a real canonical Huffman code of that text (lengths <= 12, one-level table), every stream is decoded and compared with
// the text.  Literals only: the bases and qualities of sequencing reads are literals in zlib's output too (matches are the headers and
// barcode lines), and a match costs the decoding lane a second table look-up before the wave can copy.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/inflate_rate.hip -o tools/probe/_bin/inflate_rate && tools/probe/_bin/inflate_rate
// Prints, for S concurrent streams: text MB/s per stream and GB/s in all.  The reference's files are one gzip member each
// (lib/tada/src/multifastq.rs:69-127 reads them with one decoder per file): S = the number of files of a lane, dozens.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <queue>
#include <string>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TB = 12;                 // table bits = longest code
constexpr int ORING = 8192;            // output ring (bytes): one 512-byte input chunk decodes to at most 4096 symbols

__global__ void __launch_bounds__(64) decode_kernel(const uint32_t* __restrict__ in, uint64_t in_words /* per stream, multiple of 128 */,
                                                   uint8_t* __restrict__ out, uint64_t out_cap /* per stream, multiple of 8 */,
                                                   const uint16_t* __restrict__ table_g, uint64_t n_sym) {
    __shared__ uint16_t tab[1 << TB];
    __shared__ uint32_t ibuf[2][128];
    __shared__ __attribute__((aligned(8))) uint8_t obuf[ORING];
    __shared__ uint64_t s_opos;
    const int lane = threadIdx.x;
    for (int i = lane; i < (1 << TB); i += 64) tab[i] = table_g[i];
    const uint32_t* src = in + (uint64_t)blockIdx.x * in_words;
    uint64_t* dst = reinterpret_cast<uint64_t*>(out + (uint64_t)blockIdx.x * out_cap);
    const uint64_t n_chunks = in_words / 128;
    // chunk 0 into LDS, chunk 1 into registers
    ibuf[0][lane] = src[lane]; ibuf[0][64 + lane] = src[64 + lane];
    uint32_t n0 = 0, n1 = 0;
    if (n_chunks > 1) { n0 = src[128 + lane]; n1 = src[128 + 64 + lane]; }
    __syncthreads();
    uint64_t bitbuf = 0; uint32_t nbits = 0;      // lane 0's
    uint64_t opos = 0, done = 0, flushed = 0;     // bytes produced / symbols decoded (lane 0's), bytes flushed (uniform)
    for (uint64_t c = 0; c < n_chunks; ++c) {
        if (lane == 0) {
            const uint32_t* ib = ibuf[c & 1];
            uint32_t iw = 0;
            // decode while the chunk has words left; the last bits of a chunk are decoded with the next one
            while (done < n_sym) {
                if (nbits <= 32) {
                    if (iw == 128) break;
                    bitbuf |= (uint64_t)ib[iw++] << nbits;
                    nbits += 32;
                }
                // (>= 33 bits here: two symbols per refill check, as inflate_fast / libdeflate do)
                uint32_t e = tab[bitbuf & ((1u << TB) - 1)];
                uint32_t l = e & 15u;
                obuf[opos & (ORING - 1)] = (uint8_t)(e >> 4);
                bitbuf >>= l; nbits -= l; ++opos; ++done;
                if (done < n_sym) {
                    e = tab[bitbuf & ((1u << TB) - 1)];
                    l = e & 15u;
                    obuf[opos & (ORING - 1)] = (uint8_t)(e >> 4);
                    bitbuf >>= l; nbits -= l; ++opos; ++done;
                }
            }
            s_opos = opos;
        }
        __syncthreads();
        // the next chunk goes to LDS, the one after it is asked for
        if (c + 1 < n_chunks) { ibuf[(c + 1) & 1][lane] = n0; ibuf[(c + 1) & 1][64 + lane] = n1; }
        if (c + 2 < n_chunks) { n0 = src[(c + 2) * 128 + lane]; n1 = src[(c + 2) * 128 + 64 + lane]; }
        // whole 8-byte words of output leave, coalesced
        const uint64_t upto = s_opos & ~7ull;
        for (uint64_t p = flushed + 8ull * lane; p < upto; p += 512) dst[p >> 3] = *reinterpret_cast<const uint64_t*>(&obuf[p & (ORING - 1)]);
        flushed = upto;
        __syncthreads();
    }
    if (lane == 0) for (uint64_t p = flushed; p < opos; ++p) reinterpret_cast<uint8_t*>(dst)[p] = obuf[p & (ORING - 1)];
}

static std::string make_text(size_t pairs, uint32_t seed) {
    std::string t;
    uint64_t s = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); };
    for (size_t p = 0; p < pairs; ++p) {
        t += "@SYN:" + std::to_string(1000000 + p) + "\n";
        for (int m = 0; m < 2; ++m) {
            for (int i = 0; i < 150; ++i) t += "ACGT"[rnd() & 3];
            t += '\n';
            for (int i = 0; i < 150; ++i) { const uint32_t r = rnd() % 100; t += (char)(33 + (r < 70 ? 37 + (r & 3) : r < 90 ? 25 + (r % 12) : 2 + (r % 20))); }
            t += '\n';
        }
        for (int i = 0; i < 16; ++i) t += "ACGT"[rnd() & 3];
        t += "-1\nFFFFFFFFFFFFFFFF\nACGTACGT\nFFFFFFFF\n";
    }
    return t;
}

int main(int argc, char** argv) {
    const size_t pairs = argc > 1 ? (size_t)atol(argv[1]) : 3000;          // ~2 MB of text per stream
    const std::string text = make_text(pairs, 7);
    const uint64_t n = text.size();
    // Huffman code lengths (rare symbols are floored at 2^-10 of the text so that no code is longer than TB)
    uint64_t freq[256] = {0};
    for (unsigned char ch : text) ++freq[ch];
    for (int i = 0; i < 256; ++i) if (freq[i] && freq[i] < n / 1024 + 1) freq[i] = n / 1024 + 1;
    struct node { uint64_t f; int l, r, sym; };
    std::vector<node> nodes;
    typedef std::pair<uint64_t, int> qe;
    std::priority_queue<qe, std::vector<qe>, std::greater<qe>> pq;
    for (int i = 0; i < 256; ++i) if (freq[i]) { nodes.push_back({freq[i], -1, -1, i}); pq.push({freq[i], (int)nodes.size() - 1}); }
    while (pq.size() > 1) {
        qe a = pq.top(); pq.pop(); qe b = pq.top(); pq.pop();
        nodes.push_back({a.first + b.first, a.second, b.second, -1});
        pq.push({a.first + b.first, (int)nodes.size() - 1});
    }
    int len[256] = {0};
    std::vector<std::pair<int, int>> st{{(int)nodes.size() - 1, 0}};
    while (!st.empty()) {
        auto [x, d] = st.back(); st.pop_back();
        if (nodes[x].sym >= 0) len[nodes[x].sym] = d ? d : 1;
        else { st.push_back({nodes[x].l, d + 1}); st.push_back({nodes[x].r, d + 1}); }
    }
    int maxlen = 0; for (int i = 0; i < 256; ++i) maxlen = std::max(maxlen, len[i]);
    if (maxlen > TB) { fprintf(stderr, "code length %d > %d\n", maxlen, TB); return 1; }
    // canonical codes (RFC 1951 3.2.2), stored bit-reversed: the first code bit is the stream's next bit
    uint32_t code[256] = {0}, next_code[TB + 2] = {0}, bl_count[TB + 2] = {0};
    for (int i = 0; i < 256; ++i) if (len[i]) ++bl_count[len[i]];
    { uint32_t c = 0; for (int b = 1; b <= TB; ++b) { c = (c + bl_count[b - 1]) << 1; next_code[b] = c; } }
    for (int i = 0; i < 256; ++i) if (len[i]) {
        uint32_t c = next_code[len[i]]++, r = 0;
        for (int b = 0; b < len[i]; ++b) r |= ((c >> b) & 1u) << (len[i] - 1 - b);
        code[i] = r;
    }
    std::vector<uint16_t> table(1 << TB, 0);
    for (int i = 0; i < 256; ++i) if (len[i]) for (uint32_t x = code[i]; x < (1u << TB); x += 1u << len[i]) table[x] = (uint16_t)((i << 4) | len[i]);
    // encode
    std::vector<uint32_t> words;
    { uint64_t bb = 0; int nb = 0;
      for (unsigned char ch : text) { bb |= (uint64_t)code[ch] << nb; nb += len[ch]; if (nb >= 32) { words.push_back((uint32_t)bb); bb >>= 32; nb -= 32; } }
      words.push_back((uint32_t)bb); words.push_back(0); }
    while (words.size() % 128) words.push_back(0);
    const uint64_t in_words = words.size(), out_cap = (n + 7 + 512) & ~7ull;
    double bits = 0; for (unsigned char ch : text) bits += len[ch];
    printf("text %.2f MB per stream, %.2f bits per symbol (literal-only Huffman; zlib -6 on such text: ~2.3 bits per byte with its matches), longest code %d\n",
           n / 1e6, bits / n, maxlen);
    const int S_MAX = 4096;
    uint32_t* d_in; uint8_t* d_out; uint16_t* d_tab;
    CK(hipMalloc(&d_in, (size_t)S_MAX * in_words * 4));
    CK(hipMalloc(&d_out, (size_t)S_MAX * out_cap));
    CK(hipMalloc(&d_tab, table.size() * 2));
    CK(hipMemcpy(d_tab, table.data(), table.size() * 2, hipMemcpyHostToDevice));
    for (int s = 0; s < S_MAX; ++s) CK(hipMemcpy(d_in + (size_t)s * in_words, words.data(), in_words * 4, s ? hipMemcpyHostToDevice : hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<uint8_t> back(n);
    const int Ss[] = {1, 16, 48, 64, 128, 256, 512, 1024, 2048, 4096};
    for (int S : Ss) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(d_out, 0, (size_t)S * out_cap));
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(decode_kernel, dim3(S), dim3(64), 0, 0, d_in, in_words, d_out, out_cap, d_tab, n);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            best = std::min(best, ms);
        }
        bool ok = true;
        for (int s : {0, S / 2, S - 1}) { CK(hipMemcpy(back.data(), d_out + (size_t)s * out_cap, n, hipMemcpyDeviceToHost)); ok = ok && memcmp(back.data(), text.data(), n) == 0; }
        printf("streams %5d  %9.2f ms  %8.1f MB/s of text per stream  %8.2f GB/s in all  %s\n", S, best, n / 1e6 / (best * 1e-3), (double)S * n / 1e9 / (best * 1e-3), ok ? "decoded == text" : "MISMATCH");
    }
    return 0;
}
