// Per-kernel timing with HIP events recorded on the context's stream, around the
// launches the caller selects by name.  Used by bench.py to time the dominant
// kernel inside the timed region and to print the per-kernel table.
#include "common.h"

#include <string>
#include <vector>

namespace {
struct Rec {
    int name;
    hipEvent_t e0, e1;
};
struct Prof {
    std::vector<std::string> names;
    std::vector<double> sum_ms;
    std::vector<long> count;
    std::vector<int> stream;   // 0 main, 1 detection + matching, 2 RANSAC search, 3 RANSAC polish
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
};
Prof* get(rtvm_ctx* c) {
    if (!c->prof) c->prof = new Prof;
    return (Prof*)c->prof;
}
hipEvent_t take_event(Prof* p) {
    if (!p->pool.empty()) {
        hipEvent_t e = p->pool.back();
        p->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

int rtvm_prof_begin(rtvm_ctx* c, const char* name) {
    if (c->prof_mode == 2 && !strstr(name, c->prof_filter)) return -1;
    Prof* p = get(c);
    int idx = -1;
    for (size_t i = 0; i < p->names.size(); i++)
        if (p->names[i] == name) idx = (int)i;
    if (idx < 0) {
        idx = (int)p->names.size();
        p->names.push_back(name);
        p->sum_ms.push_back(0);
        p->count.push_back(0);
        p->stream.push_back(0);
    }
    p->stream[idx] = (c->stream == c->stream_front || c->stream == c->stream_front2) ? 1 : c->stream == c->stream_rs ? 2 : c->stream == c->stream_lm ? 3 : 0;
    Rec r{idx, take_event(p), take_event(p)};
    if (!r.e0 || !r.e1) return -1;
    (void)hipEventRecord(r.e0, c->stream);
    p->recs.push_back(r);
    return (int)p->recs.size() - 1;
}

void rtvm_prof_end(rtvm_ctx* c, int rec) {
    Prof* p = get(c);
    (void)hipEventRecord(p->recs[rec].e1, c->stream);
}

void rtvm_prof_free(rtvm_ctx* c) {
    if (!c->prof) return;
    Prof* p = (Prof*)c->prof;
    for (auto& r : p->recs) {
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    for (auto e : p->pool) (void)hipEventDestroy(e);
    delete p;
    c->prof = nullptr;
}

// filter == NULL or "": time every kernel; otherwise only kernels whose name contains filter.
extern "C" int rtvm_profile_start(rtvm_ctx* c, const char* filter) {
    if (!c) return RTVM_E_ARG;
    Prof* p = get(c);
    for (size_t i = 0; i < p->sum_ms.size(); i++) p->sum_ms[i] = 0, p->count[i] = 0;
    if (filter && filter[0]) {
        snprintf(c->prof_filter, sizeof(c->prof_filter), "%s", filter);
        c->prof_mode = 2;
    } else {
        c->prof_mode = 1;
    }
    return RTVM_OK;
}

// Waits for the stream, folds the recorded intervals into the per-name totals and
// stops timing.
extern "C" int rtvm_profile_stop(rtvm_ctx* c) {
    if (!c) return RTVM_E_ARG;
    c->prof_mode = 0;
    HIP_TRY(c, hipStreamSynchronize(c->stream_front));
    HIP_TRY(c, hipStreamSynchronize(c->stream_front2));
    HIP_TRY(c, hipStreamSynchronize(c->stream_rs));
    HIP_TRY(c, hipStreamSynchronize(c->stream_lm));
    HIP_TRY(c, hipStreamSynchronize(c->stream_main));
    Prof* p = get(c);
    for (auto& r : p->recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            p->sum_ms[r.name] += ms;
            p->count[r.name] += 1;
        }
        p->pool.push_back(r.e0);
        p->pool.push_back(r.e1);
    }
    p->recs.clear();
    return RTVM_OK;
}

// One line per kernel: "name count total_ms stream" (stream: 0 = main, 1 = detection +
// matching, 2 = RANSAC search, 3 = RANSAC polish; the stream of the kernel's last launch).  Returns the number of
// bytes needed.
extern "C" int rtvm_profile_report(rtvm_ctx* c, char* buf, int buflen) {
    if (!c) return RTVM_E_ARG;
    Prof* p = get(c);
    std::string out;
    char line[256];
    for (size_t i = 0; i < p->names.size(); i++) {
        if (!p->count[i]) continue;
        snprintf(line, sizeof(line), "%s %ld %.6f %d\n", p->names[i].c_str(), p->count[i], p->sum_ms[i], p->stream[i]);
        out += line;
    }
    if (buf && buflen > 0) {
        snprintf(buf, buflen, "%s", out.c_str());
    }
    return (int)out.size() + 1;
}
