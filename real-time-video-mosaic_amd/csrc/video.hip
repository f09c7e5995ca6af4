// The reference's main loop (main.py:1596-1634: one process_frame call per frame, in order) behind the C ABI, for
// hosts that have a whole video or a stream of frames to stitch and want the batched device pipeline without
// writing its protocol themselves.  Host code only: it drives the batch entry points of stream.hip.
//
//   rtvm_process_video[_source]   frames resident on the device, the reference's rules (skip, validate, smooth,
//                                 chain) applied on the device (k_batch_glue): detection, matching and RANSAC run up
//                                 to four batches ahead of the canvas half, which follows in pieces; frames that
//                                 come back pending are dealt with here (see below);
//   rtvm_pipe_*                   process_frame as a call that only takes the frame (host memory), the rules applied
//                                 by the caller's function, frame by frame in order, between RANSAC and the warp.
//
// Why frames come back pending, and what is done about it:
//   * a blend cleared a canvas pixel (two non-zero values blended to zero): the canvas masks predicted for the
//     frames behind it in its piece are wrong.  Their detection, matching and RANSAC results stand; their canvas
//     half is queued again, one frame at first (such frames come in runs) and twice as many after every piece that
//     goes through whole; the pieces queued behind the bad one did nothing and are queued again too;
//   * the reference skipped a frame (fewer than 4 matches, or no homography; main.py:722-731): the frames
//     detected behind it were matched against it, not against the frame the reference would have used.  They are
//     released and detected again, matched against the last frame that was stitched.
#include "common.h"

#include <chrono>
#include <deque>
#include <memory>
#include <vector>

namespace {

#define TRYV(expr)                 \
    do {                           \
        const int rc_ = (expr);    \
        if (rc_ != RTVM_OK) return rc_; \
    } while (0)

double now_seconds() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

constexpr int SKIP_BITS = RTVM_FRAME_FEW_MATCHES | RTVM_FRAME_NO_HOMOGRAPHY;

// ------------------------------------------------------------------ device rules: rtvm_process_video
struct DetRec {       // a detected batch
    long ticket;      // of rtvm_batch_detect
    int first;        // video frame of its first frame
    int n;            // frames
    int queued;       // frames whose canvas half has been queued
    int consumed;     // frames consumed (stitched or skipped)
};
struct Piece {        // a canvas half in flight
    DetRec* e;
    long ticket;
    int off, cnt;     // frames off .. off+cnt-1 of the batch
};

struct VideoRun {
    rtvm_ctx* c;
    int32_t* status;
    int n;
    rtvm_video_stats st;
    rtvm_video_event* events;
    int events_cap, n_events;

    void event(int first, int count, int done, int reason) {
        if (events && n_events < events_cap) events[n_events] = rtvm_video_event{now_seconds(), first, count, done, reason};
        n_events++;
    }

    int status_of(long ticket, int cnt, int* done, int* reason, int* st_out) {
        return rtvm_batch_status_of(c, ticket, done, reason, st_out, nullptr, nullptr, cnt);
    }

    // Frames first .. k-1 of the detected batch `ticket` (video frames start + first ..) came back pending although
    // their detection results stand: their canvas half again, one frame at first, twice as many after every retry
    // that goes through whole.  The next piece is queued before the outcome of the current one is read, so the
    // device does not wait for the host between pieces; when the current one leaves frames pending, that next
    // piece did nothing and is queued again.  *redo = -1 when all k frames are consumed, else the video frame from
    // which detection has to start again (a skipped frame invalidated the matching behind it).
    int stitch_rest(long ticket, int start, int k, int first, int* redo) {
        struct Retry {
            long ticket;
            int p0, cnt;
        };
        std::deque<Retry> q;
        int m = 1, pos = first;
        int st_buf[RTVM_MAX_BATCH];
        *redo = -1;
        while (first < k) {
            while ((int)q.size() < 2 && pos < k) {
                const int cnt = std::min(m, k - pos);
                TRYV(rtvm_batch_retry(c, ticket, pos, cnt));
                q.push_back(Retry{rtvm_batch_ticket(c), pos, cnt});
                st.retries += 1;
                st.frames_retried += cnt;
                pos += cnt;
                m = std::min(2 * m, k);
            }
            const Retry r = q.front();
            q.pop_front();
            int d2 = 0, reason = 0;
            TRYV(status_of(r.ticket, r.cnt, &d2, &reason, st_buf));
            event(start + r.p0, r.cnt, d2, reason);
            for (int j = 0; j < d2; j++) status[start + r.p0 + j] = st_buf[j];
            first = r.p0 + d2;
            const bool skipped = d2 > 0 && (st_buf[d2 - 1] & SKIP_BITS);
            if (d2 == r.cnt && !skipped) continue;
            for (const Retry& v : q) {   // queued behind a piece that left frames pending: void
                int dv, rv;
                TRYV(status_of(v.ticket, v.cnt, &dv, &rv, st_buf));
                st.frames_void += v.cnt;
            }
            q.clear();
            if (skipped && d2 == r.cnt) {
                *redo = start + first < n ? start + first : -1;
                return RTVM_OK;
            }
            if (reason != RTVM_PENDING_MASK) {
                *redo = start + first;
                return RTVM_OK;
            }
            m = 1;
            pos = first;
        }
        return RTVM_OK;
    }

    int run(rtvm_frame_source source, void* user, int batch, int canvas_batch, int in_flight) {
        TRYV(rtvm_reserve_batch(c, batch));
        const int cb = std::max(1, std::min(batch, canvas_batch > 0 ? canvas_batch : (batch + 1) / 2));
        std::vector<std::unique_ptr<DetRec>> det;   // detected batches, oldest first
        std::deque<Piece> inb;                      // canvas halves in flight, oldest first
        // whatever way this function is left (a failing frame source, a HIP error), no slot set stays taken
        struct ReleaseAll {
            rtvm_ctx* c;
            std::vector<std::unique_ptr<DetRec>>& det;
            ~ReleaseAll() {
                for (auto& r : det) (void)rtvm_batch_release(c, r->ticket);
            }
        } release_all{c, det};
        int nxt = 0;                                // next frame to detect
        const uint8_t* ptrs[RTVM_MAX_BATCH];
        int st_buf[RTVM_MAX_BATCH];

        auto forget = [&](DetRec* e) {
            for (size_t i = 0; i < det.size(); i++)
                if (det[i].get() == e) {
                    det.erase(det.begin() + i);
                    return;
                }
        };
        // a skipped frame: everything detected behind it is void
        auto start_over = [&](int frame) -> int {
            for (auto& r : det) TRYV(rtvm_batch_release(c, r->ticket));
            st.frames_detected_again += nxt - frame;
            det.clear();
            inb.clear();
            nxt = frame;
            return RTVM_OK;
        };
        // the pieces queued behind a piece that left frames pending did nothing
        auto drain = [&]() -> int {
            for (const Piece& p : inb) {
                int dv, rv;
                TRYV(status_of(p.ticket, p.cnt, &dv, &rv, st_buf));
                st.frames_void += p.cnt;
            }
            inb.clear();
            for (auto& r : det) r->queued = r->consumed;
            return RTVM_OK;
        };

        while (nxt < n || !det.empty()) {
            while ((int)det.size() < RTVM_NSETS && nxt < n) {
                const int k = std::min(batch, n - nxt);
                const int rc = source(user, nxt, k, ptrs);
                if (rc != RTVM_OK) RTVM_FAIL(c, rc < 0 ? rc : RTVM_E_ARG, "process_video: the frame source failed at frame %d (%d)", nxt, rc);
                TRYV(rtvm_batch_detect(c, ptrs, k));
                det.emplace_back(new DetRec{rtvm_batch_ticket(c), nxt, k, 0, 0});
                st.batches_queued += 1;
                st.frames_queued += k;
                nxt += k;
            }
            while ((int)inb.size() < in_flight) {
                DetRec* e = nullptr;
                for (auto& r : det)
                    if (r->queued < r->n) {
                        e = r.get();
                        break;
                    }
                if (!e) break;
                const int rem = e->n - e->queued;
                // no short piece at the end of a batch: it costs a piece's fixed time
                const int cnt = rem <= cb + cb / 2 ? rem : cb;
                TRYV(rtvm_batch_retry(c, e->ticket, e->queued, cnt));
                inb.push_back(Piece{e, rtvm_batch_ticket(c), e->queued, cnt});
                e->queued += cnt;
                st.canvas_pieces += 1;
                st.canvas_frames += cnt;
            }
            const Piece p = inb.front();
            inb.pop_front();
            DetRec* e = p.e;
            const int start = e->first;
            int done = 0, reason = 0;
            TRYV(status_of(p.ticket, p.cnt, &done, &reason, st_buf));
            for (int j = 0; j < done; j++) status[start + p.off + j] = st_buf[j];
            event(start + p.off, p.cnt, done, reason);
            e->consumed = p.off + done;
            const bool skipped = done > 0 && (st_buf[done - 1] & SKIP_BITS);
            if (done == p.cnt && !skipped) {
                if (e->consumed == e->n) {
                    TRYV(rtvm_batch_release(c, e->ticket));
                    forget(e);
                }
                continue;
            }
            if (done == p.cnt) {   // a skipped last frame: what follows was matched against it
                TRYV(drain());
                if (start + e->consumed < n) {
                    TRYV(start_over(start + e->consumed));
                } else {
                    TRYV(rtvm_batch_release(c, e->ticket));
                    forget(e);
                }
                continue;
            }
            st.batches_with_pending_frames += 1;
            TRYV(drain());
            if (reason != RTVM_PENDING_MASK) {
                TRYV(start_over(start + e->consumed));
                continue;
            }
            // canvas half again for the rest of this batch; the batches behind it follow in pieces as before
            int redo = -1;
            const long ticket = e->ticket;
            TRYV(stitch_rest(ticket, start, e->n, e->consumed, &redo));
            TRYV(rtvm_batch_release(c, ticket));
            forget(e);
            if (redo >= 0) TRYV(start_over(redo));
        }
        return RTVM_OK;
    }
};

int list_source(void* user, int first, int count, const uint8_t** dev) {
    const uint8_t* const* all = static_cast<const uint8_t* const*>(user);
    for (int j = 0; j < count; j++) dev[j] = all[first + j];
    return RTVM_OK;
}

// ------------------------------------------------------------------ host rules: rtvm_pipe_*
struct PipeBatch {
    long ticket;
    long idx0;            // stream index of its first frame
    int n;                // frames (shrinks when a frame is skipped)
    bool have_res = false;
    int glued = 0;        // frames the caller's rules have been applied to
    std::vector<double> H;   // absolute homographies of the frames to stitch: frames 0 .. H.size()/9 - 1
    int queued = 0;       // of those, queued in commit pieces
    int done = 0;         // of those, confirmed stitched
    int nH() const { return (int)(H.size() / 9); }
};
struct PipePiece {
    long ticket;
    PipeBatch* b;
    int first, cnt;
};

struct Pipe {
    rtvm_ctx* c;
    rtvm_glue_fn glue;
    void* user;
    int B, P, R;
    int in_flight_max = 3;
    long n_in = 0;            // frames taken so far
    long retired = 0;         // frames whose ring slot may be written again
    std::deque<long> pend;    // stream indices in the ring, not yet detected
    std::deque<std::unique_ptr<PipeBatch>> batches;   // detected batches, oldest first
    std::deque<PipePiece> inflight;                   // commit pieces, oldest first
    int m;                    // frames of the next piece
    bool need_rebase = false, flushing = false;
    bool in_glue = false;     // inside the caller's function: put and flush are refused there
    bool failed = false;      // a call failed half way (the caller's function, a HIP error): only rtvm_pipe_close is left
    std::unique_ptr<PipeBatch> held;   // the batch of the newest frame, kept for kp_cur / matches
    long last_ticket = 0;     // batch and index of the last frame the rules were applied to
    int last_index = -1;
    rtvm_pipe_stats st{};

    int live_sets() const { return (int)batches.size() + (held ? 1 : 0); }

    int detect(bool force, bool* progressed) {
        while (!pend.empty() && ((int)pend.size() >= B || flushing || force)) {
            if (need_rebase) {
                // what is detected next is matched against the last stitched frame: its commit (which makes its
                // features the previous ones) has to be confirmed first
                bool busy = !inflight.empty();
                for (auto& b : batches) busy |= b->done < b->nH() || b->glued < b->n;
                if (busy) break;
                TRYV(rtvm_batch_rebase(c));
                need_rebase = false;
            }
            if (held) {   // the lazy reads of the last flush are over
                TRYV(rtvm_batch_release(c, held->ticket));
                held.reset();
            }
            if (live_sets() >= RTVM_NSETS) break;
            const int k = std::min(B, (int)pend.size());
            const long idx0 = pend.front();
            const uint8_t* ptrs[RTVM_MAX_BATCH];
            const int s0 = (int)(idx0 % R);
            if (s0 + k <= R) {
                TRYV(rtvm_ingest_submit_range(c, s0, k, ptrs));
            } else {
                for (int j = 0; j < k; j++) TRYV(rtvm_ingest_submit(c, (int)((idx0 + j) % R), &ptrs[j]));
            }
            for (int j = 0; j < k; j++) pend.pop_front();
            TRYV(rtvm_batch_detect(c, ptrs, k));
            std::unique_ptr<PipeBatch> b(new PipeBatch);
            b->ticket = rtvm_batch_ticket(c);
            b->idx0 = idx0;
            b->n = k;
            batches.push_back(std::move(b));
            st.batches += 1;
            *progressed = true;
            force = false;
        }
        return RTVM_OK;
    }

    int results(bool block, bool* progressed) {
        PipeBatch* b = nullptr;
        for (auto& x : batches)
            if (!x->have_res) {
                b = x.get();
                break;
            }
        if (!b) return RTVM_OK;
        if (!block) {
            const int ready = rtvm_batch_results_ready(c, b->ticket);
            if (ready < 0) return ready;
            if (!ready) return RTVM_OK;
        }
        int nm[RTVM_MAX_BATCH], fd[RTVM_MAX_BATCH];
        double Hrel[RTVM_MAX_BATCH * 9];
        TRYV(rtvm_batch_results(c, b->ticket, nm, fd, nullptr, Hrel, b->n));
        b->have_res = true;
        *progressed = true;
        for (int j = 0; j < b->n; j++) {
            double Habs[9];
            in_glue = true;
            const int keep = glue(user, b->idx0 + j, nm[j], fd[j], Hrel + 9 * j, Habs);
            in_glue = false;
            if (keep < 0) RTVM_FAIL(c, RTVM_E_CALLBACK, "pipe: the caller's rules failed at frame %ld (%d)", b->idx0 + j, keep);
            b->glued = j + 1;
            last_ticket = b->ticket;
            last_index = j;
            if (keep) {
                b->H.insert(b->H.end(), Habs, Habs + 9);
                continue;
            }
            // the frame is skipped and the state did not advance: everything detected behind it was matched
            // against the wrong frame
            long again = b->n - (j + 1);
            std::vector<long> redo;
            for (int i = j + 1; i < b->n; i++) redo.push_back(b->idx0 + i);
            b->n = j + 1;
            while (batches.back().get() != b) {
                PipeBatch* later = batches.back().get();
                TRYV(rtvm_batch_release(c, later->ticket));
                for (int i = later->n - 1; i >= 0; i--) pend.push_front(later->idx0 + i);
                again += later->n;
                batches.pop_back();
            }
            for (size_t i = redo.size(); i-- > 0;) pend.push_front(redo[i]);
            st.frames_detected_again += again;
            need_rebase = true;
            break;
        }
        return RTVM_OK;
    }

    int commit(bool* progressed) {
        while ((int)inflight.size() < in_flight_max) {
            PipeBatch* b = nullptr;
            bool blocked = false;
            for (auto& x : batches) {
                if (x->queued < x->nH()) {
                    b = x.get();
                    break;
                }
                // frame order across batches: nothing of a later batch before this one's rules have run
                if (!x->have_res) {
                    blocked = true;
                    break;
                }
            }
            if (!b || blocked) break;
            const int cnt = std::min(m, b->nH() - b->queued);
            TRYV(rtvm_batch_commit(c, b->ticket, b->queued, cnt, b->H.data() + 9 * (size_t)b->queued));
            inflight.push_back(PipePiece{rtvm_batch_ticket(c), b, b->queued, cnt});
            b->queued += cnt;
            m = std::min(2 * m, P);
            st.pieces += 1;
            *progressed = true;
        }
        return RTVM_OK;
    }

    int confirm(bool block, bool* progressed) {
        if (inflight.empty()) return RTVM_OK;
        const PipePiece p = inflight.front();
        if (!block) {
            const int ready = rtvm_batch_done(c, p.ticket);
            if (ready < 0) return ready;
            if (!ready) return RTVM_OK;
        }
        inflight.pop_front();
        int done = 0;
        TRYV(rtvm_batch_status_of(c, p.ticket, &done, nullptr, nullptr, nullptr, nullptr, 0));
        p.b->done = p.first + done;
        if (done < p.cnt) {
            // a blend cleared a canvas pixel: the frames behind it are committed again (their homographies
            // stand); the pieces queued behind did nothing
            st.pieces_with_pending_frames += 1;
            for (const PipePiece& v : inflight) {
                int dv = 0;
                TRYV(rtvm_batch_status_of(c, v.ticket, &dv, nullptr, nullptr, nullptr, nullptr, 0));
                st.pieces_void += 1;
            }
            inflight.clear();
            for (auto& x : batches) x->queued = x->done;
            m = 1;
        }
        *progressed = true;
        return RTVM_OK;
    }

    int retire(bool* progressed) {
        while (!batches.empty()) {
            PipeBatch* b = batches.front().get();
            if (!b->have_res || b->glued < b->n || b->done < b->nH()) break;
            retired = b->idx0 + b->n;
            std::unique_ptr<PipeBatch> own = std::move(batches.front());
            batches.pop_front();
            if (batches.empty() && pend.empty()) {
                held = std::move(own);   // the batch of the newest frame: kp_cur / matches are read from it
            } else {
                TRYV(rtvm_batch_release(c, own->ticket));
            }
            *progressed = true;
        }
        return RTVM_OK;
    }

    int pump(bool block) {
        bool progressed = false;
        TRYV(detect(false, &progressed));
        TRYV(results(block && inflight.empty(), &progressed));
        TRYV(commit(&progressed));
        TRYV(confirm(block, &progressed));
        TRYV(retire(&progressed));
        if (block && !progressed) {
            // nothing was there to wait for: detection must be possible now
            bool detected = false;
            TRYV(detect(true, &detected));
            if (!detected && (!pend.empty() || !batches.empty()))
                RTVM_FAIL(c, RTVM_E_STATE, "pipe stalled (%d frames waiting, %d batches)", (int)pend.size(), (int)batches.size());
        }
        return RTVM_OK;
    }

    bool busy() const { return !pend.empty() || !batches.empty(); }
};

Pipe* pipe_of(rtvm_ctx* c) { return static_cast<Pipe*>(c->pipe); }

}  // namespace

void rtvm_pipe_free(rtvm_ctx* c) {
    delete pipe_of(c);
    c->pipe = nullptr;
}

extern "C" int rtvm_process_video_source(rtvm_ctx* c, rtvm_frame_source source, void* user, int n, int batch, int canvas_batch,
                                         int pieces_in_flight, int32_t* status, rtvm_video_stats* stats,
                                         rtvm_video_event* events, int events_cap, int* n_events) {
    if (!c) return RTVM_E_ARG;
    if (!source || n < 0 || (n > 0 && !status) || batch < 1 || batch > RTVM_MAX_BATCH || events_cap < 0 || (events_cap > 0 && !events))
        RTVM_FAIL(c, RTVM_E_ARG, "process_video: bad argument (batch 1..%d)", RTVM_MAX_BATCH);
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "process_video: call rtvm_stream_begin first");
    VideoRun run{c, status, n, rtvm_video_stats{}, events, events_cap, 0};
    run.st.frames = n;
    for (int i = 0; i < n; i++) status[i] = 0;
    const int rc = run.run(source, user, batch, canvas_batch, pieces_in_flight > 0 ? pieces_in_flight : 2);
    if (stats) *stats = run.st;
    if (n_events) *n_events = run.n_events;
    return rc;
}

extern "C" int rtvm_process_video(rtvm_ctx* c, const uint8_t* const* bgr_dev, int n, int batch, int canvas_batch,
                                  int32_t* status, rtvm_video_stats* stats) {
    if (!c) return RTVM_E_ARG;
    if (n > 0 && !bgr_dev) RTVM_FAIL(c, RTVM_E_ARG, "process_video: null frame list");
    return rtvm_process_video_source(c, list_source, const_cast<uint8_t**>(bgr_dev), n, batch, canvas_batch, 0, status, stats,
                                     nullptr, 0, nullptr);
}

extern "C" int rtvm_pipe_open(rtvm_ctx* c, int batch, int piece, rtvm_glue_fn glue, void* user) {
    if (!c) return RTVM_E_ARG;
    if (!glue || batch < 1 || batch > RTVM_MAX_BATCH) RTVM_FAIL(c, RTVM_E_ARG, "pipe_open: bad argument (batch 1..%d)", RTVM_MAX_BATCH);
    if (c->pipe) RTVM_FAIL(c, RTVM_E_STATE, "pipe_open: the context has a pipe already");
    if (!c->stream_ready) RTVM_FAIL(c, RTVM_E_STATE, "pipe_open: call rtvm_first_frame or rtvm_stream_begin first");
    TRYV(rtvm_reserve_batch(c, batch));
    const int R = 6 * batch;   // four batches alive, one being filled, one spare
    if (c->ingest_slots == 0) {
        TRYV(rtvm_ingest_reserve(c, R));
    } else if (c->ingest_slots < R) {
        RTVM_FAIL(c, RTVM_E_STATE, "pipe_open: the ingest ring has %d slots, batches of %d need %d", c->ingest_slots, batch, R);
    }
    Pipe* p = new Pipe;
    p->c = c;
    p->glue = glue;
    p->user = user;
    p->B = batch;
    p->P = std::max(1, std::min(piece > 0 ? piece : std::max(1, batch / 2), batch));
    p->R = R;
    p->m = p->P;
    c->pipe = p;
    return RTVM_OK;
}

extern "C" int rtvm_pipe_put(rtvm_ctx* c, const uint8_t* bgr_host) {
    if (!c) return RTVM_E_ARG;
    Pipe* p = pipe_of(c);
    if (!p) RTVM_FAIL(c, RTVM_E_STATE, "pipe_put: no pipe (rtvm_pipe_open)");
    if (!bgr_host) RTVM_FAIL(c, RTVM_E_ARG, "pipe_put: null frame");
    if (p->in_glue) RTVM_FAIL(c, RTVM_E_STATE, "pipe_put: called from inside the rules function");
    if (p->failed) RTVM_FAIL(c, RTVM_E_STATE, "pipe_put: an earlier call on this pipe failed; close it (rtvm_pipe_close)");
    const long idx = p->n_in;
    int rc = RTVM_OK;
    while (rc == RTVM_OK && idx - p->retired >= p->R) rc = p->pump(true);   // the slot still holds a frame that may be needed again
    if (rc == RTVM_OK) rc = rtvm_ingest_write(c, (int)(idx % p->R), bgr_host);   // complete on return
    if (rc == RTVM_OK) {
        p->pend.push_back(idx);
        p->n_in += 1;
        p->st.frames += 1;
        rc = p->pump(false);
    }
    if (rc != RTVM_OK) p->failed = true;
    return rc;
}

extern "C" int rtvm_pipe_flush(rtvm_ctx* c) {
    if (!c) return RTVM_E_ARG;
    Pipe* p = pipe_of(c);
    if (!p) return RTVM_OK;
    if (p->in_glue) RTVM_FAIL(c, RTVM_E_STATE, "pipe_flush: called from inside the rules function");
    if (p->failed) RTVM_FAIL(c, RTVM_E_STATE, "pipe_flush: an earlier call on this pipe failed; close it (rtvm_pipe_close)");
    p->flushing = true;
    int rc = RTVM_OK;
    while (rc == RTVM_OK && p->busy()) rc = p->pump(true);
    p->flushing = false;
    if (rc != RTVM_OK) p->failed = true;
    return rc;
}

extern "C" int rtvm_pipe_busy(rtvm_ctx* c) {
    if (!c) return RTVM_E_ARG;
    Pipe* p = pipe_of(c);
    return p && p->busy() ? 1 : 0;
}

extern "C" int rtvm_pipe_last(rtvm_ctx* c, long* ticket, int* index) {
    if (!c) return RTVM_E_ARG;
    if (!ticket || !index) RTVM_FAIL(c, RTVM_E_ARG, "pipe_last: null argument");
    Pipe* p = pipe_of(c);
    *ticket = 0;
    *index = -1;
    if (p && p->held && p->last_ticket == p->held->ticket) {
        *ticket = p->last_ticket;
        *index = p->last_index;
    }
    return RTVM_OK;
}

extern "C" int rtvm_pipe_get_stats(rtvm_ctx* c, rtvm_pipe_stats* out) {
    if (!c) return RTVM_E_ARG;
    if (!out) RTVM_FAIL(c, RTVM_E_ARG, "pipe_get_stats: null argument");
    Pipe* p = pipe_of(c);
    *out = p ? p->st : rtvm_pipe_stats{};
    return RTVM_OK;
}

extern "C" int rtvm_pipe_close(rtvm_ctx* c) {
    if (!c) return RTVM_E_ARG;
    Pipe* p = pipe_of(c);
    if (!p) return RTVM_OK;
    // what is still held or in flight is dropped: the slot sets become free again
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream_main));
    for (auto& b : p->batches) (void)rtvm_batch_release(c, b->ticket);
    if (p->held) (void)rtvm_batch_release(c, p->held->ticket);
    rtvm_pipe_free(c);
    return RTVM_OK;
}
