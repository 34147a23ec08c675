// quadtree.inl -- data-parallel restatement of ORBextractor::DistributeOctTree
// (reference: src/ORBextractor.cc:518-721, DivideNode :467-516).
//
// The reference walks a std::list sequentially.  The same result is obtained level-synchronously:
//   * key order inside a node never matters except for "first maximum response wins", so keys
//     are never moved: each key only remembers the pool slot of the node that owns it, and the
//     winner of a node is an atomicMax over (response, -original index);
//   * one pass of the reference's outer loop splits every node holding >1 key; children are
//     push_front'ed in visiting order, so the new list is  reverse(push order) ++ (unsplit nodes
//     in their old order): two prefix sums give every node its new list position;
//   * the reference's inner "largest first" loop (sort by (size, pointer), split until the list
//     has N nodes) becomes: rank-sort the candidates, count children of all of them, prefix-sum
//     the growth in processing order, find the break index, apply only the processed ones.
// Pointer ties in the reference's sort are allocator-dependent; like the oracle this code breaks
// them by creation (push) order: later-created first.
//
// The code is written as a sequence of PHASES.  All state that lives across a phase boundary is
// in (shared) memory or is a uniform function of it, so the identical text runs
//   * on the GPU:   QT_PHASE = the calling thread, QT_SYNC = __syncthreads()
//   * in tests/:    QT_PHASE = a loop over all QT_NT logical threads, QT_SYNC = nothing
// (tests/emul builds the second form to check the formulation against the oracle on CPU; it is
// never part of the product library).
#ifndef VIEO_QUADTREE_INL
#define VIEO_QUADTREE_INL

#include <stdint.h>

#ifdef QT_DEVICE
#define QT_FN __device__ __forceinline__
#define QT_MEM __device__ __forceinline__
#define QT_NT ((int)blockDim.x)
#define QT_PHASE for (int tid = (int)threadIdx.x, qt_once_ = 1; qt_once_; qt_once_ = 0)
#define QT_SYNC() __syncthreads()
#define QT_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define QT_ATOMIC_MAX(p, v) atomicMax((p), (v))
#else
#define QT_FN static inline
#define QT_MEM inline
#define QT_NT 256
#define QT_PHASE for (int tid = 0; tid < QT_NT; ++tid)
#define QT_SYNC() ((void)0)
static inline int qt_host_add_(int* p, int v) {
  int o = *p;
  *p = o + v;
  return o;
}
static inline unsigned qt_host_max_(unsigned* p, unsigned v) {
  unsigned o = *p;
  if (v > o) *p = v;
  return o;
}
#define QT_ATOMIC_ADD(p, v) qt_host_add_((p), (v))
#define QT_ATOMIC_MAX(p, v) qt_host_max_((p), (v))
#endif

// packed candidate key: x (12 bit) | y (12 bit) << 12 | response (8 bit) << 24, coordinates
// relative to (minBorderX, minBorderY) as in vToDistributeKeys.
#define QT_KEY_X(k) ((int)((k)&0xFFFu))
#define QT_KEY_Y(k) ((int)(((k) >> 12) & 0xFFFu))
#define QT_KEY_R(k) ((int)((k) >> 24))

struct QtShared {
  // uniform scalars (written by one logical thread inside a phase, read after the sync)
  int K;        // number of candidate keys
  int n;        // current list length
  int nfresh;   // next never-used pool slot
  int E;        // expandable children created by the last round (= candidate count)
  int P, R;     // pushes / retained nodes of the round being applied
  int jstar;    // phase-2 break index
  int error;    // capacity overflow (cannot happen for NCAP >= max(N,4*nIni)+8)
};

// Pointers into (shared) memory, all sized by the caller:
struct QtMem {
  QtShared* s;
  // node pool, indexed by slot [ncap]
  short *x0, *y0, *x1, *y1;
  int* cnt;
  int* cc;               // [ncap*4] child key counts of a node being split
  unsigned short* child; // [ncap*4] pool slot of child q
  unsigned short* mark;  // [ncap]   phase 2: 1 + processing position of a candidate slot, else 0
  unsigned* best;        // [ncap]
  // list, ping-pong [2][ncap]
  unsigned short* list[2];
  // per list position / per candidate scratch
  unsigned long long* scanA;  // [scap]
  unsigned long long* scanB;  // [scap]
  unsigned char* flag;        // [ncap]
  // candidates [2][ncap]: slot and size, in push order (ping-pong: next round's are written
  // while this round's are read)
  unsigned short* cand_slot[2];
  int* cand_size[2];
  unsigned short* order;  // [ncap] processing order -> candidate index
  int ncap, scap;
  // the ping-pong halves by (uniform) index WITHOUT indexing the pointer arrays dynamically: a dynamic index forces
  // the whole struct into scratch memory and every use of a pointer becomes a scratch load inside the round loops
  QT_MEM unsigned short* list_(int i) const { return i ? list[1] : list[0]; }
  QT_MEM unsigned short* cslot(int i) const { return i ? cand_slot[1] : cand_slot[0]; }
  QT_MEM int* csize(int i) const { return i ? cand_size[1] : cand_size[0]; }
};

// Inclusive scan of a[0..n) (4 x 16-bit packed counters); result ends in the returned buffer (a or b).
#ifdef QT_DEVICE
// Device form: every thread scans a contiguous chunk, the chunk totals are scanned inside the wavefront on DPP (the two
// 32-bit halves separately: the counters are 16 bits wide and never carry into one another), the wavefront totals go
// through LDS.  Two barriers whatever n is; the Hillis-Steele form below took log2(n) of them (ten at n = 540) and was
// two thirds of the barriers of a quadtree workgroup.
QT_FN unsigned long long* qt_scan(unsigned long long* a, unsigned long long* b, int n) {
  __shared__ unsigned long long s_qt_tot[16];
  const int tid = (int)threadIdx.x, NT = (int)blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + NT - 1) / NT, i0 = tid * per;
  unsigned long long run = 0;
  for (int k = 0; k < per; k++)
    if (i0 + k < n) {
      run += a[i0 + k];
      b[i0 + k] = run;
    }
  int lo = (int)(unsigned)run, hi = (int)(unsigned)(run >> 32);
#define QT_DPP_SCAN(x)                                               \
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);     \
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);     \
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);     \
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);     \
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);    \
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  QT_DPP_SCAN(lo)
  QT_DPP_SCAN(hi)
#undef QT_DPP_SCAN
  const unsigned long long inc = (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32);
  if (lane == 63) s_qt_tot[wave] = inc;
  __syncthreads();
  // exclusive over the wavefront's chunks (per half: no borrow between them)
  unsigned long long off = (unsigned long long)(unsigned)((unsigned)inc - (unsigned)run) |
        ((unsigned long long)(unsigned)((unsigned)(inc >> 32) - (unsigned)(run >> 32)) << 32);
  for (int w = 0; w < wave; w++) {
    const unsigned long long t = s_qt_tot[w];
    off = (unsigned long long)(unsigned)((unsigned)off + (unsigned)t) |
          ((unsigned long long)(unsigned)((unsigned)(off >> 32) + (unsigned)(t >> 32)) << 32);
  }
  for (int k = 0; k < per; k++)
    if (i0 + k < n) {
      const unsigned long long v = b[i0 + k];
      b[i0 + k] = (unsigned long long)(unsigned)((unsigned)v + (unsigned)off) |
                  ((unsigned long long)(unsigned)((unsigned)(v >> 32) + (unsigned)(off >> 32)) << 32);
    }
  __syncthreads();
  return b;
}
#else
QT_FN unsigned long long* qt_scan(unsigned long long* a, unsigned long long* b, int n) {
  unsigned long long* src = a;
  unsigned long long* dst = b;
  for (int d = 1; d < n; d <<= 1) {
    QT_PHASE {
      for (int i = tid; i < n; i += QT_NT) dst[i] = src[i] + (i >= d ? src[i - d] : 0ull);
    }
    QT_SYNC();
    unsigned long long* t = src;
    src = dst;
    dst = t;
  }
  return src;
}
#endif

QT_FN int qt_quadrant(int x, int y, int x0, int y0, int x1, int y1) {
  // DivideNode: halfX = ceil((UR.x-UL.x)/2.f), halfY = ceil((BR.y-UL.y)/2.f)
  const int mx = x0 + ((x1 - x0 + 1) >> 1);
  const int my = y0 + ((y1 - y0 + 1) >> 1);
  return (x < mx) ? ((y < my) ? 0 : 2) : ((y < my) ? 1 : 3);
}

QT_FN void qt_child_rect(int q, int x0, int y0, int x1, int y1, int* cx0, int* cy0, int* cx1,
                         int* cy1) {
  const int mx = x0 + ((x1 - x0 + 1) >> 1);
  const int my = y0 + ((y1 - y0 + 1) >> 1);
  *cx0 = (q & 1) ? mx : x0;
  *cx1 = (q & 1) ? x1 : mx;
  *cy0 = (q & 2) ? my : y0;
  *cy1 = (q & 2) ? y1 : my;
}

// Apply one round: split the nodes selected by `sel` (phase 1: every listed node with >1 key;
// phase 2: candidates with processing position <= jstar).  On entry cc[] holds the child counts
// and kq[] the quadrant of every key of a node that MAY be split.  mode 0 = phase 1, 1 = phase 2.
// Returns nothing; updates list (cur -> cur^1), pool, candidates (cb -> cb^1), s->n, s->E.
QT_FN void qt_apply_round(const QtMem& m, int mode, int cur, int cb, const unsigned* keys,
                          unsigned short* kslot, unsigned char* kq) {
  QtShared* s = m.s;
  const int n = s->n;
  const unsigned short* L = m.list_(cur);
  unsigned short* Ln = m.list_(cur ^ 1);
  // ---- per list position: is the node split in this round?  packed counters:
  //  bits 0..15 pushes (children), 16..31 retained, 32..47 expandable children
  QT_PHASE {
    for (int i = tid; i < n; i += QT_NT) {
      const int sl = L[i];
      bool sp;
      if (mode == 0)
        sp = m.cnt[sl] > 1;
      else
        sp = m.mark[sl] != 0 && (int)m.mark[sl] - 1 <= s->jstar;
      m.flag[i] = sp ? 1 : 0;
      unsigned long long v;
      if (sp) {
        int nch = 0, ne = 0;
        for (int q = 0; q < 4; q++) {
          nch += m.cc[sl * 4 + q] > 0;
          ne += m.cc[sl * 4 + q] > 1;
        }
        v = (unsigned long long)nch | ((unsigned long long)ne << 32);
      } else
        v = 1ull << 16;
      // phase 2 pushes happen in PROCESSING order, not list order: handled below
      m.scanA[i] = v;
    }
  }
  QT_SYNC();
  if (mode == 0) {
    unsigned long long* inc = qt_scan(m.scanA, m.scanB, n);
    unsigned long long* oth = (inc == m.scanA) ? m.scanB : m.scanA;
    QT_PHASE {
      if (tid == 0) {
        unsigned long long tot = n > 0 ? inc[n - 1] : 0ull;
        s->P = (int)(tot & 0xFFFF);
        s->R = (int)((tot >> 16) & 0xFFFF);
        s->E = (int)((tot >> 32) & 0xFFFF);
      }
      // keep the exclusive values in `oth` so the next phase can read both safely
      for (int i = tid; i < n; i += QT_NT) oth[i] = i > 0 ? inc[i - 1] : 0ull;
    }
    QT_SYNC();
    const int P = s->P;
    const int nfresh = s->nfresh;
    if (P + s->R > m.ncap || nfresh + (P - (n - s->R)) > m.ncap) {
      QT_PHASE {
        if (tid == 0) s->error = 1;
      }
      QT_SYNC();
      return;
    }
    QT_PHASE {
      for (int i = tid; i < n; i += QT_NT) {
        const int sl = L[i];
        const unsigned long long ex = oth[i];
        const int pushbase = (int)(ex & 0xFFFF), retbase = (int)((ex >> 16) & 0xFFFF);
        const int ebase = (int)((ex >> 32) & 0xFFFF);
        if (m.flag[i]) {
          const int freshbase = pushbase - (i - retbase);  // sum over earlier split (nch-1)
          const int px0 = m.x0[sl], py0 = m.y0[sl], px1 = m.x1[sl], py1 = m.y1[sl];
          int c[4];
          for (int q = 0; q < 4; q++) c[q] = m.cc[sl * 4 + q];
          int t = 0, te = 0;
          for (int q = 0; q < 4; q++) {
            if (c[q] <= 0) continue;
            const int slot = (t == 0) ? sl : nfresh + freshbase + (t - 1);
            int cx0, cy0, cx1, cy1;
            qt_child_rect(q, px0, py0, px1, py1, &cx0, &cy0, &cx1, &cy1);
            m.x0[slot] = (short)cx0, m.y0[slot] = (short)cy0;
            m.x1[slot] = (short)cx1, m.y1[slot] = (short)cy1;
            m.cnt[slot] = c[q];
            m.child[sl * 4 + q] = (unsigned short)slot;
            Ln[P - 1 - (pushbase + t)] = (unsigned short)slot;
            if (c[q] > 1) {
              m.cslot(cb ^ 1)[ebase + te] = (unsigned short)slot;
              m.csize(cb ^ 1)[ebase + te] = c[q];
              te++;
            }
            t++;
          }
        } else {
          Ln[P + retbase] = (unsigned short)sl;
        }
      }
    }
    QT_SYNC();
  } else {
    // phase 2: pushes in processing order j = 0..jstar over candidates order[j]
    const int np = s->jstar + 1;  // processed candidates
    const int cbsz = np;
    // retained scan over list positions
    unsigned long long* inc = qt_scan(m.scanA, m.scanB, n);
    unsigned long long* oth = (inc == m.scanA) ? m.scanB : m.scanA;
    QT_PHASE {
      if (tid == 0) s->R = n > 0 ? (int)((inc[n - 1] >> 16) & 0xFFFF) : 0;
      for (int i = tid; i < n; i += QT_NT) oth[i] = i > 0 ? inc[i - 1] : 0ull;
    }
    QT_SYNC();
    // `oth` holds per-position exclusive counters; stash retbase into flag-side array via Ln
    // later.  Now scan the processed candidates in processing order (reuse `inc` buffer).
    unsigned long long* pa = inc;  // free now
    QT_PHASE {
      for (int j = tid; j < cbsz; j += QT_NT) {
        const int sl = m.cslot(cb)[m.order[j]];
        int nch = 0, ne = 0;
        for (int q = 0; q < 4; q++) {
          nch += m.cc[sl * 4 + q] > 0;
          ne += m.cc[sl * 4 + q] > 1;
        }
        pa[j] = (unsigned long long)nch | ((unsigned long long)ne << 32);
      }
    }
    QT_SYNC();
    // need a third buffer for this scan's ping-pong while `oth` must survive: use the upper
    // half of the scan arrays (scap >= 2*ncap guaranteed by the caller)
    unsigned long long* pb = pa + m.ncap;
    unsigned long long* pinc = qt_scan(pa, pb, cbsz);
    QT_PHASE {
      if (tid == 0) {
        unsigned long long tot = cbsz > 0 ? pinc[cbsz - 1] : 0ull;
        s->P = (int)(tot & 0xFFFF);
        s->E = (int)((tot >> 32) & 0xFFFF);
      }
    }
    QT_SYNC();
    const int P = s->P;
    const int nfresh = s->nfresh;
    if (P + s->R > m.ncap || nfresh + (P - np) > m.ncap) {
      QT_PHASE {
        if (tid == 0) s->error = 1;
      }
      QT_SYNC();
      return;
    }
    QT_PHASE {
      for (int j = tid; j < cbsz; j += QT_NT) {
        const int sl = m.cslot(cb)[m.order[j]];
        const unsigned long long in = pinc[j];
        int c[4];
        int nch = 0, ne = 0;
        for (int q = 0; q < 4; q++) {
          c[q] = m.cc[sl * 4 + q];
          nch += c[q] > 0;
          ne += c[q] > 1;
        }
        const int pushbase = (int)(in & 0xFFFF) - nch;
        const int ebase = (int)((in >> 32) & 0xFFFF) - ne;
        const int freshbase = pushbase - j;
        const int px0 = m.x0[sl], py0 = m.y0[sl], px1 = m.x1[sl], py1 = m.y1[sl];
        int t = 0, te = 0;
        for (int q = 0; q < 4; q++) {
          if (c[q] <= 0) continue;
          const int slot = (t == 0) ? sl : nfresh + freshbase + (t - 1);
          int cx0, cy0, cx1, cy1;
          qt_child_rect(q, px0, py0, px1, py1, &cx0, &cy0, &cx1, &cy1);
          m.x0[slot] = (short)cx0, m.y0[slot] = (short)cy0;
          m.x1[slot] = (short)cx1, m.y1[slot] = (short)cy1;
          m.cnt[slot] = c[q];
          m.child[sl * 4 + q] = (unsigned short)slot;
          Ln[P - 1 - (pushbase + t)] = (unsigned short)slot;
          if (c[q] > 1) {
            m.cslot(cb ^ 1)[ebase + te] = (unsigned short)slot;
            m.csize(cb ^ 1)[ebase + te] = c[q];
            te++;
          }
          t++;
        }
      }
      for (int i = tid; i < n; i += QT_NT) {
        if (!m.flag[i]) {
          const int retbase = (int)((oth[i] >> 16) & 0xFFFF);
          Ln[P + retbase] = L[i];
        }
      }
    }
    QT_SYNC();
  }
  // ---- move the keys of split nodes to their child slot
  const int K = s->K;
  QT_PHASE {
    for (int k = tid; k < K; k += QT_NT) {
      const int q = kq[k];
      if (q == 0xFF) continue;
      const int sl = kslot[k];
      if (mode == 1 && !(m.mark[sl] != 0 && (int)m.mark[sl] - 1 <= s->jstar)) continue;
      kslot[k] = m.child[sl * 4 + q];
    }
  }
  QT_SYNC();
  QT_PHASE {
    if (tid == 0) {
      const int np = (mode == 0) ? (n - s->R) : (s->jstar + 1);
      s->nfresh = s->nfresh + (s->P - np);
      s->n = s->P + s->R;
    }
  }
  QT_SYNC();
}

// Count children of the nodes selected for (possible) splitting: zero cc, then one pass over the
// keys.  mode 0: nodes with cnt>1;  mode 1: slots with mark != 0.
QT_FN void qt_count_children(const QtMem& m, int mode, int cur, const unsigned* keys,
                             const unsigned short* kslot, unsigned char* kq) {
  QtShared* s = m.s;
  const int n = s->n, K = s->K;
  const unsigned short* L = m.list_(cur);
  QT_PHASE {
    for (int i = tid; i < n; i += QT_NT) {
      const int sl = L[i];
      const bool sel = (mode == 0) ? (m.cnt[sl] > 1) : (m.mark[sl] != 0);
      if (sel) {
        m.cc[sl * 4 + 0] = 0, m.cc[sl * 4 + 1] = 0;
        m.cc[sl * 4 + 2] = 0, m.cc[sl * 4 + 3] = 0;
      }
    }
  }
  QT_SYNC();
  QT_PHASE {
    for (int k = tid; k < K; k += QT_NT) {
      const int sl = kslot[k];
      const bool sel = (mode == 0) ? (m.cnt[sl] > 1) : (m.mark[sl] != 0);
      if (!sel) {
        kq[k] = 0xFF;
        continue;
      }
      const unsigned key = keys[k];
      const int q =
          qt_quadrant(QT_KEY_X(key), QT_KEY_Y(key), m.x0[sl], m.y0[sl], m.x1[sl], m.y1[sl]);
      kq[k] = (unsigned char)q;
      QT_ATOMIC_ADD(&m.cc[sl * 4 + q], 1);
    }
  }
  QT_SYNC();
}

// Whole DistributeOctTree for one (image, level).  keys[0..K) are the candidates in
// vToDistributeKeys order; kslot/kq are K-sized scratch; out[] receives the selected keys in
// the reference's output (list) order.  s->K must be set and the initial-node key counts
// m.cnt[0..nIni) accumulated, with kslot[k] = initial node of key k, before the call.
// Returns the number of selected keys.
QT_FN int qt_distribute(const QtMem& m, const unsigned* keys, unsigned short* kslot,
                        unsigned char* kq, int regW, int regH, int nIni, float hX, int N,
                        unsigned* out) {
  QtShared* s = m.s;
  // ---- initial nodes (ORBextractor.cc:533-563); empty ones are erased
  QT_PHASE {
    for (int i = tid; i < nIni; i += QT_NT) {
      m.x0[i] = (short)(int)(hX * (float)i);
      m.x1[i] = (short)(int)(hX * (float)(i + 1));
      m.y0[i] = 0;
      m.y1[i] = (short)regH;
    }
    if (tid == 0) {
      int n = 0;
      for (int i = 0; i < nIni; i++)
        if (m.cnt[i] > 0) m.list[0][n++] = (unsigned short)i;
      s->n = n;
      s->nfresh = nIni;
      s->E = 0;
      s->error = 0;
    }
    for (int i = tid; i < m.ncap; i += QT_NT) m.mark[i] = 0;
  }
  QT_SYNC();
  int cur = 0, cb = 0;
  bool finish = false;
  while (!finish) {
    const int prevSize = s->n;
    qt_count_children(m, 0, cur, keys, kslot, kq);
    qt_apply_round(m, 0, cur, cb, keys, kslot, kq);
    cur ^= 1;
    cb ^= 1;
    if (s->error) break;
    const int size = s->n;
    if (size >= N || size == prevSize) {
      finish = true;
    } else if (size + s->E * 3 > N) {
      while (!finish) {
        const int prev2 = s->n;
        const int mc = s->E;  // candidates of the previous round, push order, in cand_*[cb]
        // processing order: descending (size, push index)
        QT_PHASE {
          for (int a = tid; a < mc; a += QT_NT) {
            const int sa = m.csize(cb)[a];
            int rank = 0;
            for (int b = 0; b < mc; b++) {
              const int sb = m.csize(cb)[b];
              rank += (sb > sa) || (sb == sa && b > a);
            }
            m.order[rank] = (unsigned short)a;
          }
        }
        QT_SYNC();
        QT_PHASE {
          for (int j = tid; j < mc; j += QT_NT)
            m.mark[m.cslot(cb)[m.order[j]]] = (unsigned short)(j + 1);
          if (tid == 0) s->jstar = mc - 1;
        }
        QT_SYNC();
        qt_count_children(m, 1, cur, keys, kslot, kq);
        // growth prefix in processing order -> break index
        QT_PHASE {
          for (int j = tid; j < mc; j += QT_NT) {
            const int sl = m.cslot(cb)[m.order[j]];
            int nch = 0;
            for (int q = 0; q < 4; q++) nch += m.cc[sl * 4 + q] > 0;
            m.scanA[j] = (unsigned long long)(nch - 1);
          }
        }
        QT_SYNC();
        unsigned long long* ginc = qt_scan(m.scanA, m.scanB, mc);
        QT_PHASE {
          for (int j = tid; j < mc; j += QT_NT) {
            const int after = prev2 + (int)ginc[j];
            const int before = prev2 + (j > 0 ? (int)ginc[j - 1] : 0);
            if (after >= N && before < N) s->jstar = j;
          }
        }
        QT_SYNC();
        qt_apply_round(m, 1, cur, cb, keys, kslot, kq);
        // clear marks of this round's candidates
        QT_PHASE {
          for (int j = tid; j < mc; j += QT_NT) m.mark[m.cslot(cb)[m.order[j]]] = 0;
        }
        QT_SYNC();
        cur ^= 1;
        cb ^= 1;
        if (s->error) break;
        if (s->n >= N || s->n == prev2) finish = true;
      }
    }
    if (s->error) break;
  }
  // ---- best key per node (ORBextractor.cc:697-718): max response, first in key order wins
  const int n = s->n, K = s->K;
  const unsigned short* L = m.list_(cur);
  QT_PHASE {
    for (int i = tid; i < n; i += QT_NT) m.best[L[i]] = 0u;
  }
  QT_SYNC();
  QT_PHASE {
    for (int k = tid; k < K; k += QT_NT) {
      const unsigned v = ((unsigned)QT_KEY_R(keys[k]) << 24) | (0xFFFFFFu - (unsigned)k);
      QT_ATOMIC_MAX(&m.best[kslot[k]], v);
    }
  }
  QT_SYNC();
  QT_PHASE {
    for (int i = tid; i < n; i += QT_NT) out[i] = keys[0xFFFFFFu - (m.best[L[i]] & 0xFFFFFFu)];
  }
  QT_SYNC();
  return n;
}

#endif  // VIEO_QUADTREE_INL
