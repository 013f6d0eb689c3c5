// nb_timing.h -- per-stage timing with HIP events, shared by the 2-D and 3-D handles.
//
// A step records up to MARKS events on the simulation's stream without waiting for anything.
// Every mark names the stage the time up to the NEXT mark belongs to (stages 0..5 = bbox, keys,
// sort, build, com, force; 6 = the multi-GPU exchange; STAGE_END = nothing, the mark only ends
// the stage before it), so a stage may be entered several times in a step -- the cell exchange
// has two collectives -- and the time after a step's last mark is not counted.  The host waits
// only when RING steps are pending or when the caller asks for the numbers, which are means over
// the steps since the previous query.
#pragma once

#include <utility>

#include "nb_common.h"

namespace nb {

struct StageTimer {
    static constexpr int RING = 32;   // steps whose events may be pending before the host waits
    static constexpr int MARKS = 12;  // events per step
    static constexpr int STAGES = 8;
    static constexpr int STAGE_END = -1;
    int on = 0;
    hipEvent_t ev[RING][MARKS] = {};
    signed char stage_of[RING][MARKS] = {};
    int nmark[RING] = {};  // marks recorded in each pending set
    int used = 0;          // event sets recorded and not yet read
    double sum[STAGES] = {};  // milliseconds per stage summed over `steps` steps
    int steps = 0;

    int create()
    {
        for (auto &set : ev)
            for (auto &e : set)
                if (hipEventCreate(&e) != hipSuccess)
                    return fail(NB_ERR_HIP, "hipEventCreate failed");
        return NB_OK;
    }

    void destroy()
    {
        for (auto &set : ev)
            for (auto &e : set)
                if (e) {
                    (void)hipEventDestroy(e);
                    e = nullptr;
                }
    }

    void mark(hipStream_t stream, int stage)
    {
        if (!on || used >= RING || stage >= STAGES)
            return;
        const int k = nmark[used];
        if (k >= MARKS)
            return;
        (void)hipEventRecord(ev[used][k], stream);
        stage_of[used][k] = (signed char)stage;
        nmark[used] = k + 1;
    }

    // Reads the closed event sets into the sums (waits for the stream).  A set still open -- the
    // marks of a step whose step_done has not come yet -- moves to the front of the ring.
    int drain(hipStream_t stream)
    {
        if (used == 0)
            return NB_OK;
        const int closed = used;
        used = 0;  // also when the wait fails: mark() must never index past the ring
        const hipError_t err = hipStreamSynchronize(stream);
        if (err != hipSuccess) {
            for (int &m : nmark)
                m = 0;
            return fail(NB_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(err));
        }
        for (int e = 0; e < closed; ++e) {
            for (int k = 0; k + 1 < nmark[e]; ++k) {
                float ms = 0.f;
                const int st = stage_of[e][k];
                if (st >= 0 && hipEventElapsedTime(&ms, ev[e][k], ev[e][k + 1]) == hipSuccess)
                    sum[st] += ms;
            }
            nmark[e] = 0;
            steps += 1;
        }
        if (closed < RING && nmark[closed] > 0) {
            for (int k = 0; k < MARKS; ++k) {
                std::swap(ev[0][k], ev[closed][k]);
                stage_of[0][k] = stage_of[closed][k];
            }
            nmark[0] = nmark[closed];
            nmark[closed] = 0;
        }
        return NB_OK;
    }

    // A step's events are recorded: nothing is waited for until the ring is full.
    int step_done(hipStream_t stream)
    {
        if (!on)
            return NB_OK;
        if (++used == RING)
            return drain(stream);
        return NB_OK;
    }

    void reset()
    {
        for (double &v : sum)
            v = 0.0;
        steps = 0;
    }

    int set(hipStream_t stream, int enable)
    {
        NB_TRY(drain(stream));
        on = enable ? 1 : 0;
        reset();
        return NB_OK;
    }

    // Mean milliseconds per stage since the previous call; starts a new average.
    int read(hipStream_t stream, float *out, int count)
    {
        NB_TRY(drain(stream));
        for (int k = 0; k < count; ++k)
            out[k] = (k < STAGES && steps > 0) ? (float)(sum[k] / steps) : 0.f;
        reset();
        return NB_OK;
    }
};

}  // namespace nb
