// Transformer building blocks for the TrOCR-sized recognizer:
//   layernorm_kernel        one wave64 per token row, fp32 statistics, bf16 in/out, 16-byte accesses
//   attention_fwd_kernel    flash-style attention on MFMA for full sequences (ViT encoder, 577 tokens):
//                           K/V streamed through LDS in 64-key blocks with online softmax
//   attention_decode_kernel one query per (sequence, head) against a KV cache / encoder memory;
//                           optionally appends the new K/V row at a device-side step counter, so a
//                           whole greedy-decode step has static launch parameters (hipGraph replay)
//   embed_rows_kernel       token + learned position embedding gather for the decoder
//   decode_bookkeep_kernel  tail of a greedy-decode step: record the step's token for every live sequence, sum
//                           its probability, mark sequences that emitted EOS, advance the step counter and
//                           publish "every sequence is done" - all of it inside the captured step, so the host
//                           loop around the graph replays launches nothing else
// Head dimension is 64 (768/12 for the encoder, 1024/16 for the decoder).
//
// attention_fwd orientation (per wave: 16 queries, per block: 4 waves = 64 queries):
//   S^T = K . Q^T      A = K rows (natural layout in LDS), B = Q (held in registers)
//                      -> a lane owns 4 keys x 1 query per 16-key tile: the softmax row reductions
//                         need only 2 shuffles (xor 16, 32) and the running max / sum / rescale
//                         factor are per-lane scalars (the query is the lane's column).
//   O^T = V^T . P^T    A = V^T (written to LDS transposed, keys in the order the S^T accumulators
//                         hold them), B = P^T straight from the S^T accumulator registers
//                      -> no LDS round trip for P, and a lane ends up with 4 consecutive head
//                         dims of one query -> 8-byte bf16 stores.
#include "common.h"

#define ATT_D 64
#define ATT_NEG (-1e30f)

// ------------------------------------------------------------------------------------------ layernorm
__global__ __launch_bounds__(256) void layernorm_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const bf16_t* __restrict__ res,
                                                        bf16_t* __restrict__ y, long long rows, int C, float eps) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const bf16_t* xr = x + row * C;
  const bf16_t* rr = res ? res + row * C : nullptr;
  // C <= 2048: every lane keeps its elements in registers (C/8 16-byte chunks spread over 64 lanes)
  float v[4][8];
  float s = 0.f, ss = 0.f;
  const int chunks = C >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c8 = lane + 64 * i;
    if (c8 < chunks) {
      u32x4 raw = *reinterpret_cast<const u32x4*>(xr + c8 * 8);
      u32x4 rw = {0u, 0u, 0u, 0u};
      if (rr) rw = *reinterpret_cast<const u32x4*>(rr + c8 * 8);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = bf16_bits_to_f32(raw[q] & 0xffffu) + bf16_bits_to_f32(rw[q] & 0xffffu);
        float b = bf16_bits_to_f32(raw[q] >> 16) + bf16_bits_to_f32(rw[q] >> 16);
        v[i][2 * q] = a; v[i][2 * q + 1] = b;
        s += a + b;
        ss += a * a + b * b;
      }
    }
  }
  s = wave_reduce_sum(s);
  ss = wave_reduce_sum(ss);
  const float mean = s / (float)C;
  const float var = fmaxf(ss / (float)C - mean * mean, 0.f);
  const float inv = rsqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c8 = lane + 64 * i;
    if (c8 < chunks) {
      bf16x8 o;
#pragma unroll
      for (int q = 0; q < 8; ++q) o[q] = (bf16_t)((v[i][q] - mean) * inv * gamma[c8 * 8 + q] + beta[c8 * 8 + q]);
      *reinterpret_cast<bf16x8*>(y + row * C + c8 * 8) = o;
    }
  }
}

// y = LayerNorm(x (+ res)) * gamma + beta.  x, res, y: [rows, C] bf16; C % 8 == 0, C <= 2048.
extern "C" int dlamd_layernorm(const void* x, const void* gamma, const void* beta, const void* res, void* y,
                               long long rows, int C, float eps, int reserved, hipStream_t stream) {
  (void)reserved;
  if (rows <= 0) return 0;
  if (C % 8 != 0 || C > 2048 || C <= 0) return -1;
  long long nb = (rows + 3) / 4;
  if (nb > 0x7fffffffLL) return -2;
  hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const bf16_t*)x, (const float*)gamma,
                     (const float*)beta, (const bf16_t*)res, (bf16_t*)y, rows, C, eps);
  return DLAMD_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------ attention (prefill)
struct AttArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; bf16_t* o;
  int B, H, Sq, Sk;
  long long q_bs, q_ss, k_bs, k_ss, v_bs, v_ss, o_bs, o_ss;  // batch / sequence strides in elements; head h at + h*64
  float scale;
  int causal;
};

__global__ __launch_bounds__(256) void attention_fwd_kernel(const AttArgs p) {
  // LDS: K block [64 keys][64 d] with 144-byte rows; V^T block [64 d][64 keys (permuted)] with 144-byte rows
  __shared__ __attribute__((aligned(16))) char Ks[64 * 144];
  __shared__ __attribute__((aligned(16))) char Vt[64 * 144];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, g = lane >> 4;
  const int bh = blockIdx.y;
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * 64 + wave * 16;
  const bf16_t* qb = p.q + (long long)b * p.q_bs + h * ATT_D;
  const bf16_t* kb = p.k + (long long)b * p.k_bs + h * ATT_D;
  const bf16_t* vb = p.v + (long long)b * p.v_bs + h * ATT_D;
  bf16_t* ob = p.o + (long long)b * p.o_bs + h * ATT_D;

  // Q fragment (B operand): lane holds Q[q0 + fr][d = ks*32 + 8g .. +8], pre-scaled
  const int qi = q0 + fr;
  bf16x8 qf[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    u32x4 raw = {0u, 0u, 0u, 0u};
    if (qi < p.Sq) raw = *reinterpret_cast<const u32x4*>(qb + (long long)qi * p.q_ss + ks * 32 + g * 8);
    bf16x8 t = as_bf16x8(raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (bf16_t)((float)t[j] * p.scale);
    qf[ks] = t;
  }
  f32x4 oacc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) oacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = ATT_NEG, l_run = 0.f;

  const int kend = p.causal ? min(p.Sk, blockIdx.x * 64 + 64) : p.Sk;
  for (int k0 = 0; k0 < kend; k0 += 64) {
    __syncthreads();
    // stage K (natural) and V (transposed, permuted key order): 64 rows x 128 B each, 2 x 16 B per thread
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int e = tid + it * 256;
      const int r = e >> 3, c = e & 7;           // key row, 16-byte chunk (8 dims)
      const int key = k0 + r;
      u32x4 kk = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
      if (key < p.Sk) {
        kk = *reinterpret_cast<const u32x4*>(kb + (long long)key * p.k_ss + c * 8);
        vv = *reinterpret_cast<const u32x4*>(vb + (long long)key * p.v_ss + c * 8);
      }
      *reinterpret_cast<u32x4*>(Ks + r * 144 + c * 16) = kk;
      // position of key r inside its 32-key k-step: tile t = (r>>4)&1, group gg = (r&15)>>2, j = r&3
      //   -> operand slot = 8*gg + 4*t + j   (see header comment)
      const int kstep = r >> 5, t = (r >> 4) & 1, gg = (r & 15) >> 2, j = r & 3;
      const int slot = kstep * 32 + gg * 8 + t * 4 + j;
      const bf16x8 v8 = as_bf16x8(vv);
#pragma unroll
      for (int q = 0; q < 8; ++q) *reinterpret_cast<bf16_t*>(Vt + (c * 8 + q) * 144 + slot * 2) = v8[q];
    }
    __syncthreads();
    // S^T tiles: 4 key tiles of 16 keys
    f32x4 s[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (kt * 16 + fr) * 144 + ks * 64 + g * 16);
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[kt], 0, 0, 0);
      }
    }
    // mask + online softmax (lane: keys k0 + kt*16 + 4g + j, query qi)
    float mx = ATT_NEG;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int key = k0 + kt * 16 + g * 4 + j;
        const bool ok = key < p.Sk && (!p.causal || key <= qi);
        s[kt][j] = ok ? s[kt][j] : ATT_NEG;
        mx = fmaxf(mx, s[kt][j]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);
    float psum = 0.f;
    bf16x8 pf[2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float e = s[kt][j] > 0.5f * ATT_NEG ? __expf(s[kt][j] - m_new) : 0.f;
        psum += e;
        pf[kt >> 1][(kt & 1) * 4 + j] = (bf16_t)e;
      }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) oacc[dt][j] *= alpha;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vt + (dt * 16 + fr) * 144 + ks * 64 + g * 16);
        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ks], oacc[dt], 0, 0, 0);
      }
    }
  }
  if (qi < p.Sq) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 o = {(bf16_t)(oacc[dt][0] * inv), (bf16_t)(oacc[dt][1] * inv), (bf16_t)(oacc[dt][2] * inv),
                  (bf16_t)(oacc[dt][3] * inv)};
      *reinterpret_cast<bf16x4*>(ob + (long long)qi * p.o_ss + dt * 16 + g * 4) = o;
    }
  }
}

extern "C" int dlamd_attention_fwd(const void* q, const void* k, const void* v, void* o, int B, int H, int Sq, int Sk,
                                   long long q_bs, long long q_ss, long long k_bs, long long k_ss, long long v_bs,
                                   long long v_ss, long long o_bs, long long o_ss, float scale, int causal,
                                   hipStream_t stream) {
  if (B <= 0 || H <= 0 || Sq <= 0 || Sk <= 0) return 0;
  if ((q_ss | k_ss | v_ss | q_bs | k_bs | v_bs) % 8 != 0 || (o_ss | o_bs) % 4 != 0) return -1;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15 || ((uintptr_t)o & 7)) return -2;
  if ((long long)B * H > 65535) return -3;
  AttArgs a{(const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)o, B, H, Sq, Sk,
            q_bs, q_ss, k_bs, k_ss, v_bs, v_ss, o_bs, o_ss, scale, causal};
  dim3 grid((Sq + 63) / 64, B * H);
  hipLaunchKernelGGL(attention_fwd_kernel, grid, dim3(256), 0, stream, a);
  return DLAMD_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------ attention (decode)
// One query per (b, h).  Keys/values: cache [B, Tcap, H*64] (strides given).  If knew/vnew are given, the new
// row is first appended at position *t_ptr (device-side step counter) and the query attends over [0, *t_ptr];
// otherwise it attends over klen keys (cross-attention over the encoder memory).
__global__ __launch_bounds__(256) void attention_decode_kernel(const bf16_t* __restrict__ q, long long q_bs,
                                                               bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                               long long c_bs, long long c_ss,
                                                               const bf16_t* __restrict__ knew,
                                                               const bf16_t* __restrict__ vnew, long long n_bs,
                                                               const int* __restrict__ t_ptr, int klen, int Tcap,
                                                               bf16_t* __restrict__ o, long long o_bs, int H,
                                                               float scale) {
  __shared__ float s_p[1024];
  __shared__ float s_q[ATT_D];
  __shared__ float s_red[8];
  __shared__ float s_o[4][ATT_D];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  bf16_t* kb = kc + (long long)b * c_bs + h * ATT_D;
  bf16_t* vb = vc + (long long)b * c_bs + h * ATT_D;
  int n = klen;
  if (knew) {
    const int t = min(*t_ptr, Tcap - 1);
    if (tid < ATT_D) {
      kb[(long long)t * c_ss + tid] = knew[(long long)b * n_bs + h * ATT_D + tid];
      vb[(long long)t * c_ss + tid] = vnew[(long long)b * n_bs + h * ATT_D + tid];
    }
    n = t + 1;
  }
  n = min(n, 1024);
  if (tid < ATT_D) s_q[tid] = (float)q[(long long)b * q_bs + h * ATT_D + tid] * scale;
  __syncthreads();   // also orders the append above before the cache reads below (same workgroup)
  // phase 1: 8 lanes per key (one 16-byte chunk of the 128-byte row each), a wave covers 8 keys per load instruction:
  // whole cache lines per request and 4 independent loads in flight per lane (the loop carries only the running max)
  const int sub = lane & 7, kq = lane >> 3;
  float qv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qv[j] = s_q[sub * 8 + j];
  float mx = ATT_NEG;
#pragma unroll 4
  for (int k0 = wave * 8; k0 < n; k0 += 32) {
    const int k = k0 + kq;
    float acc = 0.f;
    if (k < n) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(kb + (long long)k * c_ss + sub * 8);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        acc += bf16_bits_to_f32(raw[w] & 0xffffu) * qv[2 * w];
        acc += bf16_bits_to_f32(raw[w] >> 16) * qv[2 * w + 1];
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if (k < n) {
      if (sub == 0) s_p[k] = acc;
      mx = fmaxf(mx, acc);
    }
  }
  mx = wave_reduce_max(mx);
  if (lane == 0) s_red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  float sum = 0.f;
  for (int k = tid; k < n; k += 256) {
    float e = __expf(s_p[k] - mx);
    s_p[k] = e;
    sum += e;
  }
  sum = wave_reduce_sum(sum);
  if (lane == 0) s_red[4 + wave] = sum;
  __syncthreads();
  sum = s_red[4] + s_red[5] + s_red[6] + s_red[7];
  // phase 2: same lane layout over V rows (16-byte loads, 8 keys per wave instruction), then fold the 8 key slots
  float oacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k0 = wave * 8; k0 < n; k0 += 32) {
    const int k = k0 + kq;
    if (k < n) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(vb + (long long)k * c_ss + sub * 8);
      const float pk = s_p[k];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        oacc[2 * w] += pk * bf16_bits_to_f32(raw[w] & 0xffffu);
        oacc[2 * w + 1] += pk * bf16_bits_to_f32(raw[w] >> 16);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    oacc[j] += __shfl_xor(oacc[j], 8, 64);
    oacc[j] += __shfl_xor(oacc[j], 16, 64);
    oacc[j] += __shfl_xor(oacc[j], 32, 64);
  }
  if (kq == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s_o[wave][sub * 8 + j] = oacc[j];
  }
  __syncthreads();
  if (tid < ATT_D) {
    const float tot = s_o[0][tid] + s_o[1][tid] + s_o[2][tid] + s_o[3][tid];
    o[(long long)b * o_bs + h * ATT_D + tid] = (bf16_t)(tot / sum);
  }
}

extern "C" int dlamd_attention_decode(const void* q, long long q_bs, void* kc, void* vc, long long c_bs,
                                      long long c_ss, const void* knew, const void* vnew, long long n_bs,
                                      const void* t_ptr, int klen, int Tcap, void* o, long long o_bs, int B, int H,
                                      float scale, hipStream_t stream) {
  if (B <= 0 || H <= 0) return 0;
  if ((knew == nullptr) != (vnew == nullptr)) return -1;
  if (knew && !t_ptr) return -2;
  if (!knew && (klen <= 0 || klen > 1024)) return -3;
  if (knew && (Tcap <= 0 || Tcap > 1024)) return -4;
  if (c_ss % 8 != 0 || c_bs % 8 != 0 || ((uintptr_t)kc & 15)) return -5;
  hipLaunchKernelGGL(attention_decode_kernel, dim3(B * H), dim3(256), 0, stream, (const bf16_t*)q, q_bs, (bf16_t*)kc,
                     (bf16_t*)vc, c_bs, c_ss, (const bf16_t*)knew, (const bf16_t*)vnew, n_bs, (const int*)t_ptr, klen,
                     Tcap, (bf16_t*)o, o_bs, H, scale);
  return DLAMD_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------ embedding
// y[b, :] = tok_emb[ids[b], :] * emb_scale + pos_emb[min(*t_ptr (or t_const), P-1), :]
__global__ __launch_bounds__(256) void embed_rows_kernel(const int* __restrict__ ids, const bf16_t* __restrict__ tok,
                                                         const bf16_t* __restrict__ pos, const int* __restrict__ t_ptr,
                                                         bf16_t* __restrict__ y, long long B, int C, int V, int P,
                                                         int t_const, float emb_scale) {
  const long long b = blockIdx.x;
  int id = ids[b];
  id = max(0, min(id, V - 1));
  const int t = max(0, min(t_ptr ? *t_ptr : t_const, P - 1));
  for (int c = threadIdx.x; c < C; c += 256)
    y[b * C + c] = (bf16_t)((float)tok[(long long)id * C + c] * emb_scale + (float)pos[(long long)t * C + c]);
}

extern "C" int dlamd_embed_rows(const void* ids, const void* tok, const void* pos, const void* t_ptr, void* y,
                                long long B, int C, int V, int P, int t_const, float emb_scale, hipStream_t stream) {
  if (B <= 0) return 0;
  if (B > 0x7fffffffLL || C <= 0 || V <= 0 || P <= 0) return -1;
  hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)B), dim3(256), 0, stream, (const int*)ids, (const bf16_t*)tok,
                     (const bf16_t*)pos, (const int*)t_ptr, (bf16_t*)y, B, C, V, P, t_const, emb_scale);
  return DLAMD_CHECK_LAUNCH();
}

// ------------------------------------------------------------------------------------------ greedy-decode bookkeeping
// One workgroup.  tokens / probs: this step's argmax per sequence [B].  out: [max_len, B] token history (pre-filled
// with the pad id), psum / cnt: running probability sum and token count per sequence, done: 1 once a sequence has
// emitted eos (rows >= B_live are marked done by the caller).  t_ptr: device step counter, read as this step's index
// and incremented here.  flag: int32 in host-visible memory; set to (step + 1) the first time no sequence is live.
__global__ __launch_bounds__(256) void decode_bookkeep_kernel(const int* __restrict__ tokens, const float* __restrict__ probs,
                                                              int* __restrict__ t_ptr, int* __restrict__ out,
                                                              float* __restrict__ psum, float* __restrict__ cnt,
                                                              unsigned char* __restrict__ done, int* __restrict__ flag,
                                                              int B, int max_len, int eos) {
  __shared__ int s_live[4];
  const int t = *t_ptr;
  int live = 0;
  for (int b = threadIdx.x; b < B; b += 256) {
    if (!done[b]) {
      const int tok = tokens[b];
      if (t < max_len) {
        out[(long long)t * B + b] = tok;
        psum[b] += probs[b];
        cnt[b] += 1.f;
      }
      if (tok == eos) done[b] = 1;
      else ++live;
    }
  }
  live = (int)wave_reduce_sum((float)live);
  if ((threadIdx.x & 63) == 0) s_live[threadIdx.x >> 6] = live;
  __syncthreads();
  if (threadIdx.x == 0) {
    *t_ptr = t + 1;
    if (s_live[0] + s_live[1] + s_live[2] + s_live[3] == 0 && *flag == 0) *flag = t + 1;
  }
}

extern "C" int dlamd_decode_bookkeep(const void* tokens, const void* probs, void* t_ptr, void* out, void* psum, void* cnt,
                                     void* done, void* flag, int B, int max_len, int eos, hipStream_t stream) {
  if (B <= 0) return 0;
  if (!tokens || !probs || !t_ptr || !out || !psum || !cnt || !done || !flag || max_len <= 0) return -1;
  if (B > (1 << 22)) return -2;       // the live count goes through a float reduction
  hipLaunchKernelGGL(decode_bookkeep_kernel, dim3(1), dim3(256), 0, stream, (const int*)tokens, (const float*)probs,
                     (int*)t_ptr, (int*)out, (float*)psum, (float*)cnt, (unsigned char*)done, (int*)flag, B, max_len, eos);
  return DLAMD_CHECK_LAUNCH();
}
