/* Host-side integer loops of the batch pipeline that must consume Python's module-level `random`
 * generator draw for draw (libreco/batch/sequence.py:49-55, libreco/sampling/negatives.py:55-82).
 *
 * The generator is CPython's MT19937 (the published Matsumoto-Nishimura algorithm): the caller
 * passes `random.getstate()` (624 words + position), these functions advance it exactly as the
 * equivalent sequence of `random.randrange` / `random.random` calls would, and the caller hands
 * the state back with `random.setstate()`.  Plain C, no dependencies; built by
 * `librecommender_amd.csrc.build` into lib/liblibreco_host.so.  The Python loops in
 * batch/sequence.py and sampling/negatives.py are the definition; tests pin both to each other
 * and to the reference's fixtures.
 */
#include "../../include/libreco_host.h"

#include <string.h>

#define MT_N 624
#define MT_M 397

typedef struct {
  uint32_t* mt; /* [624] */
  int32_t pos;
} mt_gen;

static void mt_refill(mt_gen* g) {
  static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
  uint32_t* mt = g->mt;
  uint32_t y;
  int k;
  for (k = 0; k < MT_N - MT_M; ++k) {
    y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
    mt[k] = mt[k + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
  }
  for (; k < MT_N - 1; ++k) {
    y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu);
    mt[k] = mt[k + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
  }
  y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
  mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
  g->pos = 0;
}

static inline uint32_t mt_u32(mt_gen* g) {
  uint32_t y;
  if (g->pos >= MT_N) mt_refill(g);
  y = g->mt[g->pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

/* random.random(): 53-bit resolution from two outputs */
static inline double mt_double(mt_gen* g) {
  const uint32_t a = mt_u32(g) >> 5, b = mt_u32(g) >> 6;
  return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

/* random.randrange(0, n) == Random._randbelow_with_getrandbits(n) for 0 < n < 2^32 */
static inline int64_t mt_below(mt_gen* g, uint32_t n) {
  int bits = 0;
  uint32_t t = n, r;
  while (t) {
    ++bits;
    t >>= 1;
  }
  do {
    r = mt_u32(g) >> (32 - bits);
  } while (r >= n);
  return (int64_t)r;
}

/* out[j] = random.randrange(0, widths[j]); returns 0, or -1 when a width is outside (0, 2^32). */
int lrh_randrange_stream(uint32_t* mt, int32_t* pos, const int64_t* widths, int64_t count, int64_t* out) {
  mt_gen g = {mt, *pos};
  int64_t j;
  for (j = 0; j < count; ++j)
    if (widths[j] <= 0 || widths[j] >= ((int64_t)1 << 32)) return -1;
  for (j = 0; j < count; ++j) out[j] = mt_below(&g, (uint32_t)widths[j]);
  *pos = g.pos;
  return 0;
}

static int contains(const int64_t* a, int64_t lo, int64_t hi, int64_t x) {
  const int64_t end = hi;
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (a[mid] < x)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo < end && a[lo] == x;
}

/* negatives_from_unconsumed: for every (user, positive) pair draw `num_neg` items
 * n = floor(n_items * random.random()); re-draw up to `tolerance` times while n is the positive, an
 * earlier negative of the pair or consumed by the user; if that fails, up to `tolerance` more
 * times while n is the positive or an earlier negative.  `cons_ptr/cons_items` is a CSR of every
 * user's consumed items, ascending within a user.  out: [n_pairs * num_neg]. */
int lrh_negatives_unconsumed(uint32_t* mt, int32_t* pos, const int64_t* cons_ptr, const int64_t* cons_items,
                             const int64_t* users, const int64_t* items, int64_t n_pairs, int64_t n_items,
                             int32_t num_neg, int32_t tolerance, int64_t* out) {
  mt_gen g = {mt, *pos};
  int64_t p;
  for (p = 0; p < n_pairs; ++p) {
    const int64_t lo = cons_ptr[users[p]], hi = cons_ptr[users[p] + 1], item = items[p];
    int64_t* mine = out + p * num_neg;
    int32_t k;
    for (k = 0; k < num_neg; ++k) {
      int64_t n = (int64_t)((double)n_items * mt_double(&g)); /* math.floor of a non-negative double */
      int ok = 0;
      int32_t t, q;
      for (t = 0; t < tolerance; ++t) {
        int dup = 0;
        for (q = 0; q < k; ++q) dup |= (mine[q] == n);
        if (n != item && !dup && !contains(cons_items, lo, hi, n)) {
          ok = 1;
          break;
        }
        n = (int64_t)((double)n_items * mt_double(&g));
      }
      if (!ok) {
        for (t = 0; t < tolerance; ++t) {
          int dup = 0;
          for (q = 0; q < k; ++q) dup |= (mine[q] == n);
          if (n != item && !dup) break;
          n = (int64_t)((double)n_items * mt_double(&g));
        }
      }
      mine[k] = n;
    }
  }
  *pos = g.pos;
  return 0;
}

/* Pointwise collation of a feature block (batch/collators.py:PointwiseCollator._feats; reference collators.py:276-300 +
 * merge_columns :480-490) in ONE pass: sample row (r, j) — positive r, j = 0 the positive itself, j >= 1 its sampled
 * negatives — takes the batch row of positive r (all columns, original order) and replaces the ITEM columns by the
 * stored feature row of its own item.  4-byte elements (int32 sparse indices or float32 dense values).
 *   out        [n_pos * k][n_cols]
 *   batch      [n_pos][n_cols]          rows of the positives as BatchData hands them over
 *   item_rows  [n_items + 1][n_icols]   data_info.item_*_unique
 *   items      [n_pos * k]              item of every sample (positives and negatives interleaved)
 *   i_cols     [n_icols]                original column index of every item feature column                        */
int lrh_merge_pointwise_u32(uint32_t* out, const uint32_t* batch, int64_t n_pos, int k, int n_cols,
                            const uint32_t* item_rows, int64_t n_item_rows, int n_icols, const int32_t* i_cols,
                            const int64_t* items) {
  for (int t = 0; t < n_icols; ++t)
    if (i_cols[t] < 0 || i_cols[t] >= n_cols) return 1;
  for (int64_t r = 0; r < n_pos; ++r) {
    const uint32_t* src = batch + r * n_cols;
    for (int j = 0; j < k; ++j) {
      const int64_t q = r * k + j, it = items[q];
      if (it < 0 || it >= n_item_rows) return 2;
      uint32_t* dst = out + q * n_cols;
      memcpy(dst, src, (size_t)n_cols * 4);
      const uint32_t* feat = item_rows + it * n_icols;
      for (int t = 0; t < n_icols; ++t) dst[i_cols[t]] = feat[t];
    }
  }
  return 0;
}

/* out[r][:] = base[idx[r]][:] for 4-byte elements (BatchData.__getitem__: the feature rows of a batch).
 * Returns 0, or 3 for a row index out of range. */
int lrh_gather_rows_u32(uint32_t* out, const uint32_t* base, int64_t n_rows_total, const int64_t* idx, int64_t n,
                        int n_cols) {
  for (int64_t r = 0; r < n; ++r) {
    const int64_t row = idx[r];
    if (row < 0 || row >= n_rows_total) return 3;
    memcpy(out + r * n_cols, base + row * n_cols, (size_t)n_cols * 4);
  }
  return 0;
}

/* out[r][t] = t < count[r] ? (int32) hist[start[r] + t] : pad — the left-aligned, padded history windows of one batch
 * (libreco/batch/sequence.py:56-72: `seq[:length] = consumed[start:pos]`).  Returns 0, or 4 if a window leaves `hist`
 * or count[r] is outside [0, width]. */
int lrh_seq_windows_i32(int32_t* out, const int64_t* hist, int64_t hist_len, const int64_t* start, const int64_t* count,
                        int64_t n, int width, int32_t pad) {
  for (int64_t r = 0; r < n; ++r) {
    const int64_t c = count[r], s0 = start[r];
    if (c < 0 || c > width || (c > 0 && (s0 < 0 || s0 + c > hist_len))) return 4;
    int32_t* dst = out + r * width;
    const int64_t* src = hist + s0;
    int t = 0;
    for (; t < c; ++t) dst[t] = (int32_t)src[t];
    for (; t < width; ++t) dst[t] = pad;
  }
  return 0;
}

/* out[q] = position stored for the pair (users[q], items[q]) in a table of keys sorted ascending, keys[j] = user * stride +
 * item, with the keys of user u in [kptr[u], kptr[u + 1]); -1 when the pair is not in the table.  (The first position of
 * an item in a user's history: `consumed.index(item)` of libreco/batch/sequence.py:46-48.)  Returns 0, or 5 for a user
 * outside [0, n_users). */
int lrh_pair_positions(const int64_t* keys, const int64_t* first_pos, const int64_t* kptr, int64_t n_users, int64_t stride,
                       const int64_t* users, const int64_t* items, int64_t n, int64_t* out) {
  for (int64_t q = 0; q < n; ++q) {
    const int64_t u = users[q];
    if (u < 0 || u >= n_users) return 5;
    const int64_t key = u * stride + items[q];
    int64_t lo = kptr[u], hi = kptr[u + 1];
    while (lo < hi) {
      const int64_t mid = lo + ((hi - lo) >> 1);
      if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    out[q] = (lo < kptr[u + 1] && keys[lo] == key) ? first_pos[lo] : -1;
  }
  return 0;
}

int lrh_abi_version(void) { return 3; }
