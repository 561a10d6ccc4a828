// hostjoin.hip -- the host end of the decoders: label ids -> the strings the reference's decoders return.
//
// replaces: the `' '.join(self.classes[k] for k in labelling)` of BeamSearch.py:152-153 / the string loops of ctcDecoder.py:60-118 as the
// decode drivers run them once per utterance.  With three beam searches in flight the search of a 128-utterance batch costs the device
// ~0.45 ms (peaky posteriors) / ~0.8 ms (flat), and the interpreter's join over the batch's 12 k / 72 k tokens cost the host 0.4 / 2.3 ms:
// the flat regime's decode loop was host-bound.  One pass in C over the pinned result buffer: ~0.1 ms.  Host code only (no kernel, no HIP
// call): usable without a GPU, and tested there.
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "common.h"

// ctcn_join_tokens: for every row b < B the words of ids[b * row_stride + t], t < lens[b], joined by `sep` (one byte; 0: nothing between
// them) into out[out_off[b] .. out_off[b + 1]); words = the vocabulary's UTF-8 bytes back to back FOLLOWED BY 16 readable bytes, word_off[V] the
// byte offset of every word and word_len[V] its length in bytes (word_len[k] < 0: the vocabulary has no id k -- a separate array, so that a
// hole changes nothing about its neighbours; round 4 marked holes by perturbing shared offsets, which lengthened the word in front of a hole
// and made two adjacent holes cancel).  Returns the bytes written (out_off[B]), CTCN_EINVAL on bad
// arguments, CTCN_EWORKSPACE when out_cap is too small (it must hold the result + 17 bytes: nothing usable written), or -(16 + k) for an id k outside the vocabulary (the
// KeyError / IndexError of the Python expression).
extern "C" long long ctcn_join_tokens(const int32_t *ids, long long row_stride, const int32_t *lens, int B, const char *words,
                                      const int32_t *word_off, const int32_t *word_len, int V, int sep, char *out, long long out_cap,
                                      long long *out_off) {
  if (!ids || !lens || !words || !word_off || !word_len || !out || !out_off || B < 0 || V <= 0 || row_stride < 0 || out_cap < 0 || sep < 0 || sep > 255)
    return CTCN_EINVAL;
  long long pos = 0;
  for (int b = 0; b < B; ++b) {
    out_off[b] = pos;
    const int n = lens[b];
    if (n < 0 || n > row_stride) return CTCN_EINVAL;
    const int32_t *row = ids + (long long)b * row_stride;
    for (int t = 0; t < n; ++t) {
      const int k = row[t];
      if (k < 0 || k >= V || word_len[k] < 0) return -(16LL + (k < 0 ? 0x7fffffffLL : (long long)k));
      const int w0 = word_off[k], wl = word_len[k];
      if (pos + (wl > 16 ? wl : 16) + 1 > out_cap) return CTCN_EWORKSPACE;
      if (t > 0 && sep) out[pos++] = (char)sep;
      if (wl <= 16) std::memcpy(out + pos, words + w0, 16);          // (phones are a few bytes: one fixed 16-byte move -- the caller pads `words`
      else std::memcpy(out + pos, words + w0, (size_t)wl);           // with 16 bytes behind the last word -- instead of a variable-length call)
      pos += wl;
    }
  }
  out_off[B] = pos;
  return pos;
}

// ctcn_levenshtein: unit-cost edit distance of two int32 sequences (code points of two strings for the character error count, word ids for
// the word error count).  replaces: Decoder._edit_distance (ctcDecoder.py:131-150, an (L1 + 1) x (L2 + 1) list-of-lists in the interpreter,
// called twice per decoded utterance by test_ctc.py:88-95).  Two rolling rows; -1 on bad arguments.
extern "C" long long ctcn_levenshtein(const int32_t *a, long long na, const int32_t *b, long long nb) {
  if (na < 0 || nb < 0 || (na > 0 && !a) || (nb > 0 && !b)) return CTCN_EINVAL;
  if (na == 0) return nb;
  if (nb == 0) return na;
  if (nb > na) { const int32_t *t = a; a = b; b = t; const long long n = na; na = nb; nb = n; }     // (symmetric: the shorter one along the row)
  int32_t *row = static_cast<int32_t *>(malloc(sizeof(int32_t) * (size_t)(nb + 1)));
  if (!row) return CTCN_EINVAL;
  for (long long j = 0; j <= nb; ++j) row[j] = (int32_t)j;
  for (long long i = 1; i <= na; ++i) {
    const int32_t ai = a[i - 1];
    int32_t diag = row[0];                                  // dist[i-1][j-1]
    row[0] = (int32_t)i;
    int32_t left = row[0];                                  // dist[i][j-1]
    for (long long j = 1; j <= nb; ++j) {
      const int32_t up = row[j];                            // dist[i-1][j]
      int32_t best = diag + (ai != b[j - 1] ? 1 : 0);
      best = up + 1 < best ? up + 1 : best;
      best = left + 1 < best ? left + 1 : best;
      diag = up;
      row[j] = left = best;
    }
  }
  const long long d = row[nb];
  free(row);
  return d;
}
