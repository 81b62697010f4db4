// Host-side segment repair: orphan gluing and adjacent-segment merging on a segment table.
//
// Same results as the reference's tensor-mask loops (frameID/segmentation.py:91-183) but
// O(S log S): segments live in a doubly linked list and the orphans in a min-heap keyed by
// (score mean, original position), with a per-segment version stamp to retire stale entries.
// The reference picks `argsort(score_means[orphans])[0]`; equal means resolve here to the
// earliest segment.
#include <cstdint>
#include <queue>
#include <vector>

#include "cd_internal.h"

#pragma STDC FP_CONTRACT OFF

namespace {

struct Table {
    int64_t *end, *type, *runlen, *start;
    float* mean;
    std::vector<int64_t> prev, next;
    std::vector<char> alive;
};

// Segment `into` swallows its neighbour `gone` (segmentation.py:69-89).  The score-mean update
// is evaluated exactly as the reference writes it -- (a*ln + b*lg) / ln + lg, every step in
// fp32 with the run lengths converted to fp32 -- because later picks depend on its value.
void absorb(Table& t, int64_t gone, int64_t into) {
    if (gone < into) t.start[into] = t.start[gone];
    else t.end[into] = t.end[gone];
    const float ln = (float)t.runlen[into], lg = (float)t.runlen[gone];
    volatile float p1 = t.mean[into] * ln;
    volatile float p2 = t.mean[gone] * lg;
    volatile float s = p1 + p2;
    volatile float d = s / ln;
    t.mean[into] = d + lg;
    t.runlen[into] = t.end[into] - t.start[into] + 1;
}

void unlink(Table& t, int64_t i) {
    const int64_t p = t.prev[i], n = t.next[i];
    if (p >= 0) t.next[p] = n;
    if (n >= 0) t.prev[n] = p;
    t.alive[i] = 0;
}

bool orphan(const Table& t, int64_t i, int64_t k1, int64_t kb) {
    return t.runlen[i] < (t.type[i] == 2 ? kb : k1);
}

struct Entry {
    float mean;
    int64_t idx;
    uint32_t version;
    bool operator>(const Entry& o) const { return mean > o.mean || (mean == o.mean && idx > o.idx); }
};

}  // namespace

extern "C" int cd_glue_segments(int64_t* end, int64_t* type, int64_t* runlen, int64_t* start, float* mean,
                                int64_t* n_seg, int64_t real_threshold, int64_t blank_threshold, int steps) {
    if (!end || !type || !runlen || !start || !mean || !n_seg || *n_seg < 0) {
        cd::set_error("cd_glue_segments: null argument");
        return CD_EINVAL;
    }
    const int64_t S = *n_seg;
    Table t{end, type, runlen, start, mean, {}, {}, {}};
    t.prev.resize(S); t.next.resize(S); t.alive.assign(S, 1);
    for (int64_t i = 0; i < S; ++i) { t.prev[i] = i - 1; t.next[i] = i + 1 < S ? i + 1 : -1; }
    int64_t head = S > 0 ? 0 : -1, live = S;

    if (steps & 1) {
        std::vector<uint32_t> version(S, 0);
        std::priority_queue<Entry, std::vector<Entry>, std::greater<Entry>> heap;
        for (int64_t i = 0; i < S; ++i)
            if (orphan(t, i, real_threshold, blank_threshold)) heap.push(Entry{mean[i], i, 0});
        while (!heap.empty()) {
            const Entry e = heap.top();
            heap.pop();
            const int64_t i = e.idx;
            if (!t.alive[i] || version[i] != e.version) continue;
            const int64_t p = t.prev[i], n = t.next[i];
            int64_t into;
            if (p < 0) {
                if (n < 0) {   // the reference indexes row 1 of a one-row table here
                    if (steps & 4) break;          // opt-in guard: keep the lone segment
                    cd::set_error("index out of bounds: one orphan segment left, nothing to glue it to");
                    return CD_EINVAL;
                }
                into = n;
            } else if (n < 0) into = p;
            else into = t.runlen[p] > t.runlen[n] ? p : n;
            absorb(t, i, into);
            unlink(t, i);
            if (i == head) head = n;
            --live;
            ++version[into];
            if (orphan(t, into, real_threshold, blank_threshold)) heap.push(Entry{mean[into], into, version[into]});
        }
    }
    if (steps & 2) {
        for (int64_t i = head; i >= 0;) {
            const int64_t n = t.next[i];
            if (n >= 0 && t.type[i] == t.type[n]) {
                absorb(t, i, n);
                unlink(t, i);
                if (i == head) head = n;
                --live;
            }
            i = n;
        }
    }
    int64_t w = 0;
    for (int64_t i = head; i >= 0; i = t.next[i], ++w) {
        end[w] = end[i]; type[w] = type[i]; runlen[w] = runlen[i]; start[w] = start[i]; mean[w] = mean[i];
    }
    *n_seg = live;
    return CD_OK;
}
