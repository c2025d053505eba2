/*
 * fgnn_oracle.c -- CPU restatement of the SamGraph hot path.  TEST INFRASTRUCTURE ONLY
 * (see fgnn_oracle.h).  Plain C11, no dependencies; `make -C oracle` builds libfgnn_oracle.so.
 *
 * Pinning status: the RNG-free stages and the mt19937 "CPU twin" mode are checked bit-for-bit
 * against outputs of the reference's own CPU sources compiled in the build container
 * (oracle/Makefile target `_ref`, fixtures under tests/golden/, generator
 * tests/golden/make_golden.py).  The Philox mode shares every line of the sampling / dedup /
 * remap / cache / gather code with the twin mode and differs only in where a draw comes from.
 */
#include "fgnn_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ Philox4x32-10 ---------- */

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u

void fgnn_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += PHILOX_W0;
    k1 += PHILOX_W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void fgnn_philox_draw(uint64_t seed, uint64_t batch_key, uint32_t tag, uint32_t item, uint32_t block,
                      uint32_t out[4]) {
  uint32_t ctr[4] = {block, item, tag, (uint32_t)batch_key};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(batch_key >> 32)};
  fgnn_philox4x32_10(ctr, key, out);
}

uint32_t fgnn_philox_u32(uint64_t seed, uint64_t batch_key, uint32_t tag, uint32_t item, uint32_t j) {
  uint32_t r[4];
  fgnn_philox_draw(seed, batch_key, tag, item, j >> 2, r);
  return r[j & 3u];
}

/* ------------------------------------------------------------------ mt19937 ---------------- */

void fgnn_mt19937_seed(fgnn_mt19937 *g, uint32_t seed) {
  g->mt[0] = seed;
  for (int i = 1; i < 624; ++i)
    g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
  g->idx = 624;
}

uint32_t fgnn_mt19937_next(fgnn_mt19937 *g) {
  if (g->idx >= 624) {
    for (int i = 0; i < 624; ++i) {
      uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
      uint32_t v = g->mt[(i + 397) % 624] ^ (y >> 1);
      if (y & 1u) v ^= 0x9908b0dfu;
      g->mt[i] = v;
    }
    g->idx = 0;
  }
  uint32_t y = g->mt[g->idx++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

/* libstdc++-11 bits/uniform_int_dist.h: for a generator spanning exactly 32 bits the
 * distribution uses Lemire's nearly-divisionless method (_S_nd) on 64-bit products. */
uint32_t fgnn_mt19937_uniform_int(fgnn_mt19937 *g, uint32_t lo, uint32_t hi) {
  uint32_t urange = hi - lo;
  if (urange == 0xFFFFFFFFu) return fgnn_mt19937_next(g) + lo;
  uint32_t range = urange + 1u;
  uint64_t product = (uint64_t)fgnn_mt19937_next(g) * (uint64_t)range;
  uint32_t low = (uint32_t)product;
  if (low < range) {
    uint32_t threshold = (0u - range) % range;
    while (low < threshold) {
      product = (uint64_t)fgnn_mt19937_next(g) * (uint64_t)range;
      low = (uint32_t)product;
    }
  }
  return (uint32_t)(product >> 32) + lo;
}

void fgnn_rng_init(fgnn_rng *r, int mode, uint64_t seed) {
  r->mode = mode;
  r->seed = seed;
  fgnn_mt19937_seed(&r->mt, 5489u); /* default-constructed std::mt19937, cpu_random.cc:27 */
}

/* ------------------------------------------------------------------ helpers ---------------- */

size_t fgnn_predict_num_nodes(size_t batch_size, const size_t *fanout, size_t num_fanout_to_comp) {
  size_t count = batch_size;
  for (long i = (long)num_fanout_to_comp - 1; i >= 0; --i) count += count * fanout[i];
  return count;
}

size_t fgnn_table_size(size_t num, size_t scale) {
  /* 1 << (size_t)(1 + log2(num >> 1)) with the double log2 of the reference == position of the
   * highest set bit of (num >> 1), plus one. */
  size_t half = num >> 1;
  size_t lg = 0;
  while ((half >> (lg + 1)) != 0) ++lg; /* floor(log2(half)) for half >= 1 */
  size_t next_pow2 = (size_t)1 << (1 + lg);
  return next_pow2 << scale;
}

size_t fgnn_dtype_bytes(int dtype) {
  switch (dtype) {
    case FGNN_I8:
    case FGNN_U8: return 1;
    case FGNN_F16: return 2;
    case FGNN_F32:
    case FGNN_I32: return 4;
    case FGNN_I64:
    case FGNN_F64: return 8;
    default: return 4;
  }
}

static uint32_t khop_tag(int sample_type, uint32_t layer) { return ((uint32_t)sample_type << 8) | (layer & 0xffu); }

/* compaction shared by khop0 / khop2: std::remove_if(kEmptyKey) over the padded
 * [num_input x fanout] arrays (cpu_sampling_khop0.cc:69-80) == count_edge + compact_edge
 * (cuda_sampling_khop0.cu:92-174). */
static size_t compact_padded(uint32_t *src, uint32_t *dst, size_t n) {
  size_t w = 0;
  for (size_t r = 0; r < n; ++r) {
    if (src[r] != FGNN_EMPTY_KEY) {
      src[w] = src[r];
      dst[w] = dst[r];
      ++w;
    }
  }
  return w;
}

/* ------------------------------------------------------------------ khop0 ------------------ */

void fgnn_oracle_sample_khop0(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                              size_t num_input, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                              size_t *num_out, fgnn_rng *rng, uint64_t batch_key, uint32_t layer) {
  const uint32_t tag = khop_tag(FGNN_KHOP0, layer);
  for (size_t i = 0; i < num_input; ++i) {
    const uint32_t rid = input[i];
    const uint32_t off = indptr[rid];
    const uint32_t len = indptr[rid + 1] - off;
    uint32_t *s = out_src + i * fanout;
    uint32_t *d = out_dst + i * fanout;
    if (len <= fanout) {
      size_t j = 0;
      for (; j < len; ++j) { s[j] = rid; d[j] = indices[off + j]; }
      for (; j < fanout; ++j) { s[j] = FGNN_EMPTY_KEY; d[j] = FGNN_EMPTY_KEY; }
    } else {
      for (size_t j = 0; j < fanout; ++j) { s[j] = rid; d[j] = indices[off + j]; }
      for (size_t j = fanout; j < len; ++j) {
        uint32_t k;
        if (rng->mode == FGNN_RNG_MT_CPU_TWIN) {
          /* cpu_sampling_khop0.cc:61: RandomID(0, j + 1), bounds inclusive */
          k = fgnn_mt19937_uniform_int(&rng->mt, 0, (uint32_t)(j + 1));
        } else {
          /* cuda_sampling_khop0.cu:80: curand() % (j + 1) */
          k = fgnn_philox_u32(rng->seed, batch_key, tag, (uint32_t)i, (uint32_t)j) % (uint32_t)(j + 1);
        }
        if (k < fanout) d[k] = indices[off + j];
      }
    }
  }
  *num_out = compact_padded(out_src, out_dst, num_input * fanout);
}

/* ------------------------------------------------------------------ khop2 ------------------ */

void fgnn_oracle_sample_khop2(const uint32_t *indptr, uint32_t *indices, const uint32_t *input,
                              size_t num_input, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                              size_t *num_out, fgnn_rng *rng, uint64_t batch_key, uint32_t layer) {
  const uint32_t tag = khop_tag(FGNN_KHOP2, layer);
  for (size_t i = 0; i < num_input; ++i) {
    const uint32_t rid = input[i];
    const uint32_t off = indptr[rid];
    const uint32_t len = indptr[rid + 1] - off;
    uint32_t *s = out_src + i * fanout;
    uint32_t *d = out_dst + i * fanout;
    if (len <= fanout) {
      size_t j = 0;
      for (; j < len; ++j) { s[j] = rid; d[j] = indices[off + j]; }
      for (; j < fanout; ++j) { s[j] = FGNN_EMPTY_KEY; d[j] = FGNN_EMPTY_KEY; }
    } else {
      for (size_t j = 0; j < fanout; ++j) {
        uint32_t sel;
        if (rng->mode == FGNN_RNG_MT_CPU_TWIN) {
          /* cpu_sampling_khop2.cc:57: RandomID(0, len - j - 1) */
          sel = fgnn_mt19937_uniform_int(&rng->mt, 0, (uint32_t)(len - j - 1));
        } else {
          /* cuda_sampling_khop2.cu:75: curand() % (len - j) */
          sel = fgnn_philox_u32(rng->seed, batch_key, tag, (uint32_t)i, (uint32_t)j) % (uint32_t)(len - j);
        }
        const uint32_t picked = indices[off + sel];
        s[j] = rid;
        d[j] = picked;
        /* swap indices[off+sel] <-> indices[off+len-j-1] (khop2.cu:80-82) */
        indices[off + sel] = indices[off + len - j - 1];
        indices[off + len - j - 1] = picked;
      }
    }
  }
  *num_out = compact_padded(out_src, out_dst, num_input * fanout);
}

/* ------------------------------------------------------------------ weighted prefix -------- */

static float philox_uniform_float(uint32_t x) {
  /* (0,1], 24 significant bits -- stands in for curand_uniform (which is also (0,1]) */
  return (float)((x >> 8) + 1u) * (1.0f / 16777216.0f);
}

static double philox_uniform_double(uint32_t x) {
  /* (0,1) -- stands in for curand_uniform_double */
  return ((double)x + 0.5) * (1.0 / 4294967296.0);
}

typedef struct { uint32_t src, dst; size_t ord; } pair_ord;
static int cmp_pair_src_stable(const void *a, const void *b) {
  const pair_ord *x = (const pair_ord *)a, *y = (const pair_ord *)b;
  if (x->src != y->src) return x->src < y->src ? -1 : 1;
  if (x->ord != y->ord) return x->ord < y->ord ? -1 : 1;
  return 0;
}

/* post-processing shared by the three with-replacement samplers (khop1.cu:130-234, weighted_khop.cu:132-236,
 * weighted_khop_prefix.cu:148-255): stable sort by src, drop a pair equal to its successor, drop kEmptyKey */
static size_t sort_and_adjacent_dedup(pair_ord *tmp, size_t num_task, uint32_t *out_src, uint32_t *out_dst) {
  /* cub::DeviceRadixSort::SortPairs(key = src, val = dst): stable, ascending, kEmptyKey last */
  qsort(tmp, num_task, sizeof(pair_ord), cmp_pair_src_stable);
  size_t w = 0;
  for (size_t t = 0; t < num_task; ++t) {
    int keep;
    if (t + 1 < num_task)
      keep = (tmp[t].src != tmp[t + 1].src || tmp[t].dst != tmp[t + 1].dst) && tmp[t].src != FGNN_EMPTY_KEY;
    else
      keep = tmp[t].src != FGNN_EMPTY_KEY;
    if (keep) { out_src[w] = tmp[t].src; out_dst[w] = tmp[t].dst; ++w; }
  }
  return w;
}

/* mode 0: weighted prefix (binary search), 1: khop1 (uniform with replacement), 2: alias method */
static void sample_with_replacement(int mode, int sample_type, const uint32_t *indptr, const uint32_t *indices,
                                    const float *table_f, const uint32_t *alias, const uint32_t *input,
                                    size_t num_input, size_t fanout, uint32_t *out_src, uint32_t *out_dst,
                                    size_t *num_out, fgnn_rng *rng, uint64_t batch_key, uint32_t layer) {
  const uint32_t tag = khop_tag(sample_type, layer);
  const size_t num_task = num_input * fanout;
  pair_ord *tmp = (pair_ord *)malloc(sizeof(pair_ord) * (num_task ? num_task : 1));
  if (rng->mode != FGNN_RNG_PHILOX) abort(); /* no CPU twin exists in the reference (empty stubs) */
  for (size_t t = 0; t < num_task; ++t) {
    const size_t i = t / fanout, j = t % fanout;
    const uint32_t rid = input[i];
    const uint32_t off = indptr[rid];
    const uint32_t len = indptr[rid + 1] - off;
    tmp[t].ord = t;
    if (len == 0) {
      tmp[t].src = FGNN_EMPTY_KEY;
      tmp[t].dst = FGNN_EMPTY_KEY;
      continue;
    }
    uint32_t pick;
    if (mode == 1) {
      /* khop1.cu:64: curand() % len */
      pick = indices[off + fgnn_philox_u32(rng->seed, batch_key, tag, (uint32_t)i, (uint32_t)j) % len];
    } else if (mode == 2) {
      /* weighted_khop.cu:64-70: k = curand() % len; r = curand_uniform(); r < prob[k] ? indices[k] : alias[k] */
      const uint32_t k = fgnn_philox_u32(rng->seed, batch_key, tag, (uint32_t)i, 2u * (uint32_t)j) % len;
      const float r = philox_uniform_float(fgnn_philox_u32(rng->seed, batch_key, tag, (uint32_t)i, 2u * (uint32_t)j + 1u));
      pick = r < table_f[off + k] ? indices[off + k] : alias[off + k];
    } else {
      const float upbound = table_f[off + len - 1];
      const float x =
          philox_uniform_float(fgnn_philox_u32(rng->seed, batch_key, tag, (uint32_t)i, (uint32_t)j)) * upbound;
      if (x <= table_f[off]) {
        pick = indices[off];
      } else {
        size_t lo = off, hi = (size_t)off + len - 1;
        while (hi - lo >= 2) {
          size_t mid = (lo + hi) >> 1;
          if (table_f[mid] >= x) hi = mid; else lo = mid;
        }
        pick = indices[hi];
      }
    }
    tmp[t].src = rid;
    tmp[t].dst = pick;
  }
  *num_out = sort_and_adjacent_dedup(tmp, num_task, out_src, out_dst);
  free(tmp);
}

void fgnn_oracle_sample_weighted_khop_prefix(const uint32_t *indptr, const uint32_t *indices,
                                             const float *prob_prefix, const uint32_t *input,
                                             size_t num_input, size_t fanout, uint32_t *out_src,
                                             uint32_t *out_dst, size_t *num_out, fgnn_rng *rng,
                                             uint64_t batch_key, uint32_t layer) {
  sample_with_replacement(0, FGNN_WEIGHTED_KHOP_PREFIX, indptr, indices, prob_prefix, NULL, input, num_input, fanout,
                          out_src, out_dst, num_out, rng, batch_key, layer);
}

void fgnn_oracle_sample_khop1(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                              size_t num_input, size_t fanout, uint32_t *out_src, uint32_t *out_dst, size_t *num_out,
                              fgnn_rng *rng, uint64_t batch_key, uint32_t layer) {
  sample_with_replacement(1, FGNN_KHOP1, indptr, indices, NULL, NULL, input, num_input, fanout, out_src, out_dst,
                          num_out, rng, batch_key, layer);
}

void fgnn_oracle_sample_weighted_khop(const uint32_t *indptr, const uint32_t *indices, const float *prob_table,
                                      const uint32_t *alias_table, const uint32_t *input, size_t num_input,
                                      size_t fanout, uint32_t *out_src, uint32_t *out_dst, size_t *num_out,
                                      fgnn_rng *rng, uint64_t batch_key, uint32_t layer) {
  sample_with_replacement(2, FGNN_WEIGHTED_KHOP, indptr, indices, prob_table, alias_table, input, num_input, fanout,
                          out_src, out_dst, num_out, rng, batch_key, layer);
}

/* weighted_khop_hash_dedup (cuda_sampling_weighted_khop_hash_dedup.cu:41-111): alias-method draws, a draw whose
 * VALUE this seed has already selected is rejected (the per-thread 50-slot table tagged with the seed id is a set
 * membership test), until `fanout` distinct values are found; rows of length <= fanout are taken whole in CSR
 * order.  Output padded to fanout per seed, then count + compact (seed-major, no sort).
 * Attempt t of a seed uses draws 2t (k = draw % len) and 2t+1 (r = uniform (0,1]); r > prob[k] picks the alias.
 * Deviation, stated: the reference never terminates on a row with fewer than `fanout` distinct selectable values
 * (and for fanout > 50, its table size); here a seed gives up after FGNN_HASH_DEDUP_MAX_ATTEMPTS(fanout) attempts and
 * emits what it has, and fanout > FGNN_HASH_DEDUP_MAX_FANOUT is rejected by the callers. */
void fgnn_oracle_sample_weighted_khop_hash_dedup(const uint32_t *indptr, const uint32_t *indices,
                                                 const float *prob_table, const uint32_t *alias_table,
                                                 const uint32_t *input, size_t num_input, size_t fanout,
                                                 uint32_t *out_src, uint32_t *out_dst, size_t *num_out, fgnn_rng *rng,
                                                 uint64_t batch_key, uint32_t layer) {
  const uint32_t tag = khop_tag(FGNN_WEIGHTED_KHOP_HASH_DEDUP, layer);
  if (rng->mode != FGNN_RNG_PHILOX || fanout > FGNN_HASH_DEDUP_MAX_FANOUT) abort();
  for (size_t t = 0; t < num_input * fanout; ++t) out_src[t] = out_dst[t] = FGNN_EMPTY_KEY;
  for (size_t i = 0; i < num_input; ++i) {
    const uint32_t rid = input[i];
    const uint32_t off = indptr[rid];
    const uint32_t len = indptr[rid + 1] - off;
    uint32_t *src = out_src + i * fanout, *dst = out_dst + i * fanout;
    if (len <= fanout) {
      for (uint32_t j = 0; j < len; ++j) { src[j] = rid; dst[j] = indices[off + j]; }
      continue;
    }
    size_t selected = 0;
    const uint32_t max_attempts = FGNN_HASH_DEDUP_MAX_ATTEMPTS(fanout);
    for (uint32_t a = 0; a < max_attempts && selected < fanout; ++a) {
      const uint32_t k = fgnn_philox_u32(rng->seed, batch_key, tag, (uint32_t)i, 2u * a) % len;
      const float r = philox_uniform_float(fgnn_philox_u32(rng->seed, batch_key, tag, (uint32_t)i, 2u * a + 1u));
      uint32_t v = indices[off + k];
      if (r > prob_table[off + k]) v = alias_table[off + k];
      int seen = 0;
      for (size_t q = 0; q < selected; ++q) seen |= dst[q] == v;
      if (seen) continue;
      src[selected] = rid;
      dst[selected] = v;
      ++selected;
    }
  }
  *num_out = compact_padded(out_src, out_dst, num_input * fanout);
}

/* ------------------------------------------------------------------ random walk + top-K ---- */

typedef struct { uint32_t dst, count; size_t first; } visit_cnt;
static int cmp_visit(const void *a, const void *b) {
  const visit_cnt *x = (const visit_cnt *)a, *y = (const visit_cnt *)b;
  if (x->count != y->count) return x->count > y->count ? -1 : 1;
  if (x->first != y->first) return x->first < y->first ? -1 : 1;
  return 0;
}

void fgnn_oracle_sample_random_walk(const uint32_t *indptr, const uint32_t *indices,
                                    const uint32_t *input, size_t num_input, size_t walk_len,
                                    double restart_prob, size_t num_walks, size_t K,
                                    uint32_t *out_src, uint32_t *out_dst, uint32_t *out_data,
                                    size_t *num_out, fgnn_rng *rng, uint64_t batch_key,
                                    uint32_t layer) {
  const uint32_t tag = khop_tag(FGNN_RANDOM_WALK, layer);
  const size_t per_node = num_walks * walk_len;
  uint32_t *visited = (uint32_t *)malloc(sizeof(uint32_t) * (per_node ? per_node : 1));
  visit_cnt *cnt = (visit_cnt *)malloc(sizeof(visit_cnt) * (per_node ? per_node : 1));
  if (rng->mode != FGNN_RNG_PHILOX) abort();
  size_t w = 0;
  for (size_t i = 0; i < num_input; ++i) {
    const uint32_t start = input[i];
    /* layout pos = step * num_walks + walk (cuda_sampling_random_walk.cu:77-78) */
    for (size_t walk = 0; walk < num_walks; ++walk) {
      uint32_t node = start;
      for (size_t step = 0; step < walk_len; ++step) {
        const size_t pos = step * num_walks + walk;
        if (node == FGNN_EMPTY_KEY) { visited[pos] = FGNN_EMPTY_KEY; continue; }
        const uint32_t off = indptr[node];
        const uint32_t len = indptr[node + 1] - off;
        if (len == 0) { visited[pos] = FGNN_EMPTY_KEY; node = FGNN_EMPTY_KEY; continue; }
        /* two draws per step: 2d picks the neighbour, 2d+1 decides the restart */
        const uint32_t d = (uint32_t)(walk * walk_len + step);
        const uint32_t k = fgnn_philox_u32(rng->seed, batch_key, tag, (uint32_t)i, 2u * d) % len;
        node = indices[off + k];
        visited[pos] = node;
        if (philox_uniform_double(fgnn_philox_u32(rng->seed, batch_key, tag, (uint32_t)i, 2u * d + 1u)) <
            restart_prob)
          node = FGNN_EMPTY_KEY;
      }
    }
    /* FrequencyHashmap::GetTopK: distinct (seed, dst) with counts, (count desc, first pos asc) */
    size_t nd = 0;
    for (size_t p = 0; p < per_node; ++p) {
      if (visited[p] == FGNN_EMPTY_KEY) continue;
      size_t q = 0;
      for (; q < nd; ++q) if (cnt[q].dst == visited[p]) break;
      if (q == nd) { cnt[nd].dst = visited[p]; cnt[nd].count = 1; cnt[nd].first = p; ++nd; }
      else cnt[q].count++;
    }
    qsort(cnt, nd, sizeof(visit_cnt), cmp_visit);
    const size_t take = nd < K ? nd : K;
    for (size_t q = 0; q < take; ++q) {
      out_src[w] = start;
      out_dst[w] = cnt[q].dst;
      out_data[w] = cnt[q].count;
      ++w;
    }
  }
  *num_out = w;
  free(visited);
  free(cnt);
}

/* ------------------------------------------------------------------ hashtable -------------- */

fgnn_oracle_ht *fgnn_oracle_ht_create(size_t num_node, size_t capacity) {
  fgnn_oracle_ht *ht = (fgnn_oracle_ht *)calloc(1, sizeof(*ht));
  ht->o2n = (uint32_t *)malloc(sizeof(uint32_t) * (num_node ? num_node : 1));
  ht->n2o = (uint32_t *)malloc(sizeof(uint32_t) * (capacity ? capacity : 1));
  ht->num_node = num_node;
  ht->capacity = capacity;
  ht->num_items = 0;
  memset(ht->o2n, 0xFF, sizeof(uint32_t) * num_node);
  return ht;
}

void fgnn_oracle_ht_destroy(fgnn_oracle_ht *ht) {
  if (!ht) return;
  free(ht->o2n);
  free(ht->n2o);
  free(ht);
}

void fgnn_oracle_ht_reset(fgnn_oracle_ht *ht) {
  for (size_t i = 0; i < ht->num_items; ++i) ht->o2n[ht->n2o[i]] = FGNN_EMPTY_KEY;
  ht->num_items = 0;
}

int fgnn_oracle_ht_fill_unique(fgnn_oracle_ht *ht, const uint32_t *items, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    if (ht->o2n[items[i]] != FGNN_EMPTY_KEY) return -1;
    const uint32_t local = (uint32_t)(ht->num_items + i);
    ht->o2n[items[i]] = local;
    ht->n2o[local] = items[i];
  }
  ht->num_items += n;
  return 0;
}

void fgnn_oracle_ht_fill_duplicates(fgnn_oracle_ht *ht, const uint32_t *items, size_t n,
                                    uint32_t *unique, size_t *num_unique) {
  for (size_t i = 0; i < n; ++i) {
    const uint32_t id = items[i];
    if (ht->o2n[id] == FGNN_EMPTY_KEY) {
      const uint32_t local = (uint32_t)ht->num_items++;
      ht->o2n[id] = local;
      ht->n2o[local] = id;
    }
  }
  if (unique) memcpy(unique, ht->n2o, sizeof(uint32_t) * ht->num_items);
  *num_unique = ht->num_items;
}

void fgnn_oracle_map_edges(const fgnn_oracle_ht *ht, const uint32_t *src, const uint32_t *dst, size_t n,
                           uint32_t *new_src, uint32_t *new_dst) {
  for (size_t i = 0; i < n; ++i) {
    new_src[i] = ht->o2n[src[i]];
    new_dst[i] = ht->o2n[dst[i]];
  }
}

/* ------------------------------------------------------------------ batch driver ----------- */

fgnn_oracle_task *fgnn_oracle_do_sample(const uint32_t *indptr, uint32_t *indices,
                                        const float *prob_prefix, const fgnn_oracle_sample_cfg *cfg,
                                        fgnn_oracle_ht *ht, const uint32_t *seeds, size_t num_seeds,
                                        fgnn_rng *rng, uint64_t batch_key) {
  fgnn_oracle_task *task = (fgnn_oracle_task *)calloc(1, sizeof(*task));
  const size_t L = cfg->num_layers;
  task->num_layers = L;
  task->graphs = (fgnn_oracle_graph *)calloc(L ? L : 1, sizeof(fgnn_oracle_graph));

  fgnn_oracle_ht_reset(ht);
  if (fgnn_oracle_ht_fill_unique(ht, seeds, num_seeds) != 0) abort();

  uint32_t *cur_input = (uint32_t *)malloc(sizeof(uint32_t) * (num_seeds ? num_seeds : 1));
  memcpy(cur_input, seeds, sizeof(uint32_t) * num_seeds);
  size_t num_input = num_seeds;

  for (long i = (long)L - 1; i >= 0; --i) {
    const size_t fanout = cfg->fanout[i];
    const size_t cap = num_input * fanout;
    uint32_t *out_src = (uint32_t *)malloc(sizeof(uint32_t) * (cap ? cap : 1));
    uint32_t *out_dst = (uint32_t *)malloc(sizeof(uint32_t) * (cap ? cap : 1));
    uint32_t *out_data = NULL;
    size_t num_out = 0;
    switch (cfg->sample_type) {
      case FGNN_KHOP0:
        fgnn_oracle_sample_khop0(indptr, indices, cur_input, num_input, fanout, out_src, out_dst, &num_out,
                                 rng, batch_key, (uint32_t)i);
        break;
      case FGNN_KHOP2:
        fgnn_oracle_sample_khop2(indptr, indices, cur_input, num_input, fanout, out_src, out_dst, &num_out,
                                 rng, batch_key, (uint32_t)i);
        break;
      case FGNN_WEIGHTED_KHOP_PREFIX:
        fgnn_oracle_sample_weighted_khop_prefix(indptr, indices, prob_prefix, cur_input, num_input, fanout,
                                                out_src, out_dst, &num_out, rng, batch_key, (uint32_t)i);
        break;
      case FGNN_KHOP1:
        fgnn_oracle_sample_khop1(indptr, indices, cur_input, num_input, fanout, out_src, out_dst, &num_out, rng,
                                 batch_key, (uint32_t)i);
        break;
      case FGNN_WEIGHTED_KHOP:
        /* prob_prefix carries the prob table, cfg->alias_table the alias ids */
        fgnn_oracle_sample_weighted_khop(indptr, indices, prob_prefix, cfg->alias_table, cur_input, num_input, fanout,
                                         out_src, out_dst, &num_out, rng, batch_key, (uint32_t)i);
        break;
      case FGNN_WEIGHTED_KHOP_HASH_DEDUP:
        fgnn_oracle_sample_weighted_khop_hash_dedup(indptr, indices, prob_prefix, cfg->alias_table, cur_input, num_input,
                                                    fanout, out_src, out_dst, &num_out, rng, batch_key, (uint32_t)i);
        break;
      case FGNN_RANDOM_WALK:
        out_data = (uint32_t *)malloc(sizeof(uint32_t) * (cap ? cap : 1));
        fgnn_oracle_sample_random_walk(indptr, indices, cur_input, num_input, cfg->walk_len,
                                       cfg->restart_prob, cfg->num_walks, cfg->num_neighbor, out_src,
                                       out_dst, out_data, &num_out, rng, batch_key, (uint32_t)i);
        break;
      default:
        abort();
    }
    size_t num_unique = 0;
    uint32_t *unique = (uint32_t *)malloc(sizeof(uint32_t) * (num_out + ht->num_items + 1));
    fgnn_oracle_ht_fill_duplicates(ht, out_dst, num_out, unique, &num_unique);

    uint32_t *new_src = (uint32_t *)malloc(sizeof(uint32_t) * (num_out ? num_out : 1));
    uint32_t *new_dst = (uint32_t *)malloc(sizeof(uint32_t) * (num_out ? num_out : 1));
    fgnn_oracle_map_edges(ht, out_src, out_dst, num_out, new_src, new_dst);

    fgnn_oracle_graph *g = &task->graphs[i];
    g->num_src = num_unique;   /* cuda_loops.cc:211 */
    g->num_dst = num_input;    /* :212 */
    g->num_edge = num_out;     /* :213 */
    g->col = new_src;          /* :214-217 col = new_src */
    g->row = new_dst;          /* :218-221 row = new_dst */
    g->data = out_data;
    task->total_edges += num_out;

    free(out_src);
    free(out_dst);
    free(cur_input);
    cur_input = unique;
    num_input = num_unique;
  }
  task->input_nodes = cur_input;
  task->num_input_nodes = num_input;
  return task;
}

void fgnn_oracle_task_free(fgnn_oracle_task *t) {
  if (!t) return;
  for (size_t i = 0; i < t->num_layers; ++i) {
    free(t->graphs[i].row);
    free(t->graphs[i].col);
    free(t->graphs[i].data);
  }
  free(t->graphs);
  free(t->input_nodes);
  free(t);
}

/* ------------------------------------------------------------------ cache + gather --------- */

void fgnn_oracle_cache_table_build(const uint32_t *ranking_nodes, size_t num_cached, size_t num_node,
                                   uint32_t *table) {
  for (size_t i = 0; i < num_node; ++i) table[i] = FGNN_EMPTY_KEY;
  for (size_t i = 0; i < num_cached; ++i) table[ranking_nodes[i]] = (uint32_t)i;
}

void fgnn_oracle_get_miss_cache_index(const uint32_t *table, const uint32_t *nodes, size_t n,
                                      uint32_t *miss_src, uint32_t *miss_dst, size_t *num_miss,
                                      uint32_t *cache_src, uint32_t *cache_dst, size_t *num_cache) {
  size_t nm = 0, nc = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint32_t slot = table[nodes[i]];
    if (slot == FGNN_EMPTY_KEY) {
      miss_dst[nm] = (uint32_t)i;   /* position in the batch (cuda_cache.cu:102) */
      miss_src[nm] = nodes[i];      /* global node id        (cuda_cache.cu:104) */
      ++nm;
    } else {
      cache_dst[nc] = (uint32_t)i;  /* cuda_cache.cu:145 */
      cache_src[nc] = slot;         /* cache slot, cuda_cache.cu:147 */
      ++nc;
    }
  }
  *num_miss = nm;
  *num_cache = nc;
}

void fgnn_oracle_extract(void *dst, const void *src, const uint32_t *index, size_t num_index, size_t dim,
                         int dtype) {
  const size_t row = dim * fgnn_dtype_bytes(dtype);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < num_index; ++i)
    memcpy((char *)dst + i * row, (const char *)src + (size_t)index[i] * row, row);
}

/* cpu_mock_extract / CPUMockExtract, cpu/cpu_extraction.cc:44-62, 92-116 (gpu_mock_extract, cuda_extraction.cu:50-70):
 * the row id is masked to a table of 2^empty_feat_bits rows */
void fgnn_oracle_mock_extract(void *dst, const void *src, const uint32_t *index, size_t num_index, size_t dim,
                              int dtype, unsigned empty_feat_bits) {
  const size_t row = dim * fgnn_dtype_bytes(dtype);
  const size_t mask = ((size_t)1 << empty_feat_bits) - 1;
  for (size_t i = 0; i < num_index; ++i)
    memcpy((char *)dst + i * row, (const char *)src + ((size_t)index[i] & mask) * row, row);
}

void fgnn_oracle_combine(void *out, const void *rows, const uint32_t *src_index, const uint32_t *dst_index,
                         size_t n, size_t dim, int dtype) {
  const size_t row = dim * fgnn_dtype_bytes(dtype);
  for (size_t i = 0; i < n; ++i) {
    const size_t s = src_index ? src_index[i] : i;
    memcpy((char *)out + (size_t)dst_index[i] * row, (const char *)rows + s * row, row);
  }
}

static int cmp_u64_desc(const void *a, const void *b) {
  const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x > y ? -1 : (x < y ? 1 : 0);
}

void fgnn_oracle_presample_rank(const uint32_t *freq, size_t num_node, uint32_t *ranking_nodes) {
  uint64_t *t = (uint64_t *)malloc(sizeof(uint64_t) * (num_node ? num_node : 1));
  for (size_t i = 0; i < num_node; ++i) t[i] = ((uint64_t)freq[i] << 32) | (uint64_t)i;
  qsort(t, num_node, sizeof(uint64_t), cmp_u64_desc);
  for (size_t i = 0; i < num_node; ++i) ranking_nodes[i] = (uint32_t)t[i];
  free(t);
}

size_t fgnn_oracle_extract_neighbour(const uint32_t *indptr, const uint32_t *indices, const uint32_t *input,
                                     size_t num_input, uint32_t *out) {
  size_t n = 0;
  for (size_t i = 0; i < num_input; ++i) {
    const uint32_t off = indptr[input[i]], len = indptr[input[i] + 1] - off; /* count_edge, :53-58 */
    if (out) memcpy(out + n, indices + off, sizeof(uint32_t) * len);          /* compact_edge, :86-105 */
    n += len;
  }
  return n;
}

size_t fgnn_oracle_sample_all_neighbour(const uint32_t *indptr, const uint32_t *indices, const uint32_t *seeds,
                                        size_t num_seeds, size_t num_layers, size_t num_node, uint32_t *input_nodes) {
  uint8_t *seen = (uint8_t *)calloc(num_node ? num_node : 1, 1);
  size_t n = 0;
  for (size_t i = 0; i < num_seeds; ++i) { /* FillWithUnique, cuda_loops.cc:512-517 */
    seen[seeds[i]] = 1;
    input_nodes[n++] = seeds[i];
  }
  for (size_t l = 0; l < num_layers; ++l) { /* :526-565: input = unique = everything seen so far */
    const size_t num_input = n;
    for (size_t i = 0; i < num_input; ++i) {
      const uint32_t v = input_nodes[i];
      for (uint32_t e = indptr[v]; e < indptr[v + 1]; ++e) { /* GPUExtractNeighbour + FillWithDupMutable */
        const uint32_t u = indices[e];
        if (!seen[u]) {
          seen[u] = 1;
          input_nodes[n++] = u;
        }
      }
    }
  }
  free(seen);
  return n;
}

/* ------------------------------------------------------------------ shufflers -------------- */

/* std::minstd_rand0: x <- 16807 x mod (2^31 - 1); seed 0 maps to 1; min 1, max 2^31-2. */
static uint64_t minstd0_next(uint64_t *x) {
  *x = (*x * 16807ull) % 2147483647ull;
  return *x;
}

void fgnn_oracle_shuffle_minstd0(uint32_t *data, size_t n, uint64_t seed) {
  uint64_t x = seed % 2147483647ull;
  if (x == 0) x = 1;
  const uint64_t urngmin = 1, urngrange = 2147483646ull - 1ull; /* max - min */
  for (size_t i = n - 1; n > 0 && i > 0; --i) {
    /* uniform_int_distribution<size_t>(0, i)(g), libstdc++-11 bits/uniform_int_dist.h:294-359 */
    const uint64_t urange = (uint64_t)i;
    uint64_t ret;
    if (urngrange > urange) {
      const uint64_t uerange = urange + 1;
      const uint64_t scaling = urngrange / uerange;
      const uint64_t past = uerange * scaling;
      do ret = minstd0_next(&x) - urngmin; while (ret >= past);
      ret /= scaling;
    } else if (urngrange < urange) {
      /* upscaling: never reached for train sets < 2^31 items */
      abort();
    } else {
      ret = minstd0_next(&x) - urngmin;
    }
    const uint32_t tmp = data[i];
    data[i] = data[ret];
    data[ret] = tmp;
  }
}

void fgnn_oracle_dist_shuffler_partition(size_t num_data, size_t batch_size, int sampler_id,
                                         int num_sampler, size_t *dataset_offset,
                                         size_t *num_local_step, size_t *local_data_size,
                                         size_t *last_batch_size, size_t *epoch_step) {
  /* drop_last == false path of dist_shuffler.cc:47-79 */
  size_t num_step = (num_data + batch_size - 1) / batch_size;
  size_t last = num_data % batch_size == 0 ? batch_size : num_data % batch_size;
  if (sampler_id < num_sampler - 1) last = batch_size;
  *epoch_step = num_step;
  *dataset_offset = (num_step / (size_t)num_sampler * (size_t)sampler_id) * batch_size;
  if (sampler_id == num_sampler - 1) {
    const size_t previous_step = num_step / (size_t)num_sampler * (size_t)sampler_id;
    *num_local_step = num_step - previous_step;
    *local_data_size = num_data - previous_step * batch_size;
  } else {
    *num_local_step = num_step / (size_t)num_sampler;
    *local_data_size = *num_local_step * batch_size;
  }
  *last_batch_size = last;
}

void fgnn_oracle_aligned_shuffler_partition(size_t num_data, size_t batch_size, size_t worker_id,
                                            size_t num_worker, size_t *padded_size,
                                            size_t *data_per_worker, size_t *num_local_step,
                                            size_t *num_global_step, size_t *global_step_offset,
                                            size_t *global_data_offset, size_t *last_batch_size) {
  /* DistAlignedShuffler's constructor, dist_shuffler_aligned.cc:45-71: the set is rounded up to a
   * multiple of num_worker (the pad repeats its first ids, :50-59), every worker owns
   * padded / num_worker consecutive ids of the shuffled array and ceil(that / batch) steps */
  size_t padded = (num_data + num_worker - 1) / num_worker * num_worker;
  size_t per = padded / num_worker;
  size_t local = (per + batch_size - 1) / batch_size;
  *padded_size = padded;
  *data_per_worker = per;
  *num_local_step = local;
  *num_global_step = local * num_worker;
  *global_step_offset = local * worker_id;
  *global_data_offset = per * worker_id;
  *last_batch_size = per % batch_size == 0 ? batch_size : per % batch_size;
}

/* ------------------------------------------------------------------ OpenMP CPU baseline -----
 * The reference's CPU sampling path as it actually runs with omp_thread_num > 1
 * (cpu/cpu_sampling_khop2.cc:29-76 with a thread_local default-seeded mt19937 per thread,
 * cpu/cpu_random.cc:26-30; cpu/cpu_hashtable2.cc:53-194 Populate/MapNodes/MapEdges with static
 * schedules; cpu/cpu_extraction.cc:31-49).  Used ONLY as the timed cpu_baseline of bench.py: with
 * more than one thread its output depends on the thread count exactly like the reference's does
 * (every thread replays the same mt19937 stream; CAS winners are arbitrary), so it is checked for
 * invariants, not bit-exactness. */
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { uint32_t key, index, local, version; } omp_bucket;

struct fgnn_omp_ctx {
  omp_bucket *o2n;     /* [num_node], direct-indexed like CPUHashTable2 */
  uint32_t *n2o;
  size_t num_node, capacity, num_items;
  uint32_t version;
  int threads;
  fgnn_mt19937 *mt;    /* one generator per thread, all default-seeded (thread_local in the reference) */
};

fgnn_omp_ctx *fgnn_omp_create(size_t num_node, size_t capacity, int threads) {
  fgnn_omp_ctx *c = (fgnn_omp_ctx *)calloc(1, sizeof(*c));
  c->o2n = (omp_bucket *)malloc(sizeof(omp_bucket) * num_node);
  c->n2o = (uint32_t *)malloc(sizeof(uint32_t) * capacity);
  c->num_node = num_node;
  c->capacity = capacity;
  c->threads = threads > 0 ? threads : 1;
  c->mt = (fgnn_mt19937 *)malloc(sizeof(fgnn_mt19937) * (size_t)c->threads);
  for (int t = 0; t < c->threads; ++t) fgnn_mt19937_seed(&c->mt[t], 5489u);
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (size_t i = 0; i < num_node; ++i) c->o2n[i].key = FGNN_EMPTY_KEY;
  return c;
}

void fgnn_omp_destroy(fgnn_omp_ctx *c) {
  if (!c) return;
  free(c->o2n); free(c->n2o); free(c->mt); free(c);
}

static void omp_reset(fgnn_omp_ctx *c) {
#pragma omp parallel for num_threads(c->threads) schedule(static)
  for (size_t i = 0; i < c->num_items; ++i) c->o2n[c->n2o[i]].key = FGNN_EMPTY_KEY;
  c->num_items = 0;
  c->version = 0;
}

static void omp_populate(fgnn_omp_ctx *c, const uint32_t *input, size_t n) {
  const int T = c->threads;
  size_t *cnt = (size_t *)calloc((size_t)T * 8 + 8, sizeof(size_t));
#pragma omp parallel for num_threads(T) schedule(static)
  for (size_t i = 0; i < n; ++i) {
    const uint32_t id = input[i];
    if (__sync_val_compare_and_swap(&c->o2n[id].key, FGNN_EMPTY_KEY, id) == FGNN_EMPTY_KEY) {
      c->o2n[id].index = (uint32_t)i;
      c->o2n[id].version = c->version;
    }
  }
#pragma omp parallel num_threads(T)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    size_t local = 0;
#pragma omp for schedule(static)
    for (size_t i = 0; i < n; ++i) {
      const omp_bucket *b = &c->o2n[input[i]];
      if (b->index == i && b->version == c->version) ++local;
    }
    cnt[(size_t)t * 8] = local;
  }
  size_t prefix = 0;
  for (int t = 0; t < T; ++t) { const size_t v = cnt[(size_t)t * 8]; cnt[(size_t)t * 8] = prefix; prefix += v; }
  const size_t start = c->num_items;
#pragma omp parallel num_threads(T)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    size_t off = cnt[(size_t)t * 8];
#pragma omp for schedule(static)
    for (size_t i = 0; i < n; ++i) {
      omp_bucket *b = &c->o2n[input[i]];
      if (b->index == i && b->version == c->version) {
        const uint32_t id = (uint32_t)(start + off++);
        b->local = id;
        c->n2o[id] = input[i];
      }
    }
  }
  c->num_items += prefix;
  c->version++;
  free(cnt);
}

/* DoCPUSample + DoFeatureExtract for one batch (cpu/cpu_loops.cc:55-227) with khop2.
 * Returns the number of sampled edges; *num_input_nodes gets |input_nodes|.  feat_out (may be NULL)
 * receives the gathered rows, indexed through `feat_row_mask` like CPUMockExtract. */
size_t fgnn_omp_sample_batch(fgnn_omp_ctx *c, const uint32_t *indptr, uint32_t *indices, const uint32_t *seeds,
                             size_t num_seeds, const size_t *fanout, size_t num_layers, const float *feat,
                             size_t feat_dim, uint32_t feat_row_mask, float *feat_out, size_t *num_input_nodes) {
  const int T = c->threads;
  omp_reset(c);
  omp_populate(c, seeds, num_seeds);
  uint32_t *cur = (uint32_t *)malloc(sizeof(uint32_t) * (num_seeds ? num_seeds : 1));
  memcpy(cur, seeds, sizeof(uint32_t) * num_seeds);
  size_t num_input = num_seeds, total = 0;
  for (long l = (long)num_layers - 1; l >= 0; --l) {
    const size_t F = fanout[l];
    uint32_t *src = (uint32_t *)malloc(sizeof(uint32_t) * (num_input * F + 1));
    uint32_t *dst = (uint32_t *)malloc(sizeof(uint32_t) * (num_input * F + 1));
#pragma omp parallel num_threads(T)
    {
#ifdef _OPENMP
      fgnn_mt19937 *g = &c->mt[omp_get_thread_num()];
#else
      fgnn_mt19937 *g = &c->mt[0];
#endif
#pragma omp for schedule(static)
      for (size_t i = 0; i < num_input; ++i) {
        const uint32_t rid = cur[i], off = indptr[rid], len = indptr[rid + 1] - off;
        uint32_t *s = src + i * F, *d = dst + i * F;
        if (len <= F) {
          size_t j = 0;
          for (; j < len; ++j) { s[j] = rid; d[j] = indices[off + j]; }
          for (; j < F; ++j) { s[j] = FGNN_EMPTY_KEY; d[j] = FGNN_EMPTY_KEY; }
        } else {
          for (size_t j = 0; j < F; ++j) {
            const uint32_t k = fgnn_mt19937_uniform_int(g, 0, (uint32_t)(len - j - 1));
            const uint32_t v = indices[off + k];
            s[j] = rid; d[j] = v;
            indices[off + k] = indices[off + len - j - 1];
            indices[off + len - j - 1] = v;
          }
        }
      }
    }
    const size_t num_out = compact_padded(src, dst, num_input * F);  /* single-thread remove_if, khop2.cc:66-75 */
    omp_populate(c, dst, num_out);
    const size_t num_unique = c->num_items;
    uint32_t *unique = (uint32_t *)malloc(sizeof(uint32_t) * (num_unique + 1));
#pragma omp parallel for num_threads(T) schedule(static)
    for (size_t i = 0; i < num_unique; ++i) unique[i] = c->n2o[i];
    uint32_t *ns = (uint32_t *)malloc(sizeof(uint32_t) * (num_out + 1));
    uint32_t *nd = (uint32_t *)malloc(sizeof(uint32_t) * (num_out + 1));
#pragma omp parallel for num_threads(T) schedule(static)
    for (size_t i = 0; i < num_out; ++i) { ns[i] = c->o2n[src[i]].local; nd[i] = c->o2n[dst[i]].local; }
    total += num_out;
    free(src); free(dst); free(ns); free(nd); free(cur);
    cur = unique;
    num_input = num_unique;
  }
  if (feat && feat_out) {
#pragma omp parallel for num_threads(T) schedule(static)
    for (size_t i = 0; i < num_input; ++i)
      memcpy(feat_out + i * feat_dim, feat + (size_t)(cur[i] & feat_row_mask) * feat_dim, feat_dim * sizeof(float));
  }
  *num_input_nodes = num_input;
  free(cur);
  return total;
}
