// nb_timing.h -- per-stage timing with HIP events, shared by the 2-D and 3-D handles.
//
// A step records up to 8 events on the simulation's stream (stage k lies between marks k and
// k + 1; the stage after the force kernel, "other", is the multi-GPU exchange) without waiting
// for anything; the host only waits when RING steps are pending or when
// the caller asks for the numbers, which are means over the steps since the previous query.
#pragma once

#include "nb_common.h"

namespace nb {

struct StageTimer {
    static constexpr int RING = 32;  // steps whose events may be pending before the host waits
    int on = 0;
    hipEvent_t ev[RING][8] = {};
    int nmark[RING] = {};  // marks recorded in each pending set
    int used = 0;       // event sets recorded and not yet read
    double sum[8] = {};  // milliseconds per stage summed over `steps` steps
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

    void mark(hipStream_t stream, int k)
    {
        if (on && used < RING && k < 8) {
            (void)hipEventRecord(ev[used][k], stream);
            if (nmark[used] < k + 1)
                nmark[used] = k + 1;
        }
    }

    // Reads the event sets recorded so far into the sums (waits for the stream).  A direct-sum
    // step has marks 0 and 1 round its one stage, reported as stage 5, and mark 2 after the
    // exchange (stage 6); Barnes-Hut has stages 0..5 and the exchange as stage 6.
    int drain(hipStream_t stream, bool direct)
    {
        if (used == 0)
            return NB_OK;
        const int pending = used;
        used = 0;  // also when the wait fails: mark() must never index past the ring
        const hipError_t err = hipStreamSynchronize(stream);
        if (err != hipSuccess) {
            for (int &m : nmark)
                m = 0;
            return fail(NB_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(err));
        }
        for (int e = 0; e < pending; ++e) {
            for (int k = 0; k + 1 < nmark[e]; ++k) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev[e][k], ev[e][k + 1]) == hipSuccess)
                    sum[direct ? 5 + k : k] += ms;
            }
            nmark[e] = 0;
            steps += 1;
        }
        return NB_OK;
    }

    // A step's events are recorded: nothing is waited for until the ring is full.
    int step_done(hipStream_t stream, bool direct)
    {
        if (!on)
            return NB_OK;
        if (++used == RING)
            return drain(stream, direct);
        return NB_OK;
    }

    void reset()
    {
        for (double &v : sum)
            v = 0.0;
        steps = 0;
    }

    int set(hipStream_t stream, bool direct, int enable)
    {
        NB_TRY(drain(stream, direct));
        on = enable ? 1 : 0;
        reset();
        return NB_OK;
    }

    // Mean milliseconds per stage since the previous call; starts a new average.
    int read(hipStream_t stream, bool direct, float *out, int count)
    {
        NB_TRY(drain(stream, direct));
        for (int k = 0; k < count; ++k)
            out[k] = (k < 8 && steps > 0) ? (float)(sum[k] / steps) : 0.f;
        reset();
        return NB_OK;
    }
};

}  // namespace nb
