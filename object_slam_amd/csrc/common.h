// Shared host-side helpers for the gfx950 hot-path library (product code; never includes oracle/).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/oslam_hip.h"

namespace oslam {

void set_error(const char* fmt, ...);

#define OSLAM_HIP_CHECK(expr)                                                              \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            oslam::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return OSLAM_E_HIP;                                                            \
        }                                                                                  \
    } while (0)

// Waits for everything enqueued on `s`: hipStreamSynchronize (the default), or — OSLAM_WAIT_SPIN_US >= 0 — a poll of that many microseconds followed by a
// sleep on an event created with hipEventBlockingSync, which leaves the core to other threads.  Measured in bench.py's steady state (8 handles, 16 cores,
// 150-step pre-roll): spinning 20.9 k frames/s, 60 us poll + sleep 18.7 k, and neither a third thread per handle (17.5 k) nor 12 handles (15.0 k) gains from
// the freed cores: the wake-up latency of ~400 waits per step costs more than the spinning.  Kept as a knob.
hipError_t stream_wait(hipStream_t s);
// This host thread's own wait mode from now on (microseconds of polling before it sleeps on a blocking event; 0 = sleep at once, -1 = hipStreamSynchronize,
// -2 = pure polling): the local-BA service thread is off every handle's critical path and sleeps, so that its core goes to the handles' workers.
void stream_wait_thread_mode(int spin_us);
// The operator table of a driver handle calls its sub-handles one after the other from one thread: they can share that handle's stream instead of
// owning one each (streams beyond the device's hardware queues share queues, and a queue executes its packets in order whatever stream they came from).
// The caller keeps the stream alive for the lifetime of the sub-handle.
void lba_use_stream(struct ::oslam_lba* h, hipStream_t s);
// Solver handles that share a gate run the device part of their calls (upload .. download) one at a time; the window preparation stays outside the gate.
void lba_use_gate(struct ::oslam_lba* h, std::mutex* gate);
void bow_use_stream(struct ::oslam_bow* h, hipStream_t s);
void mappoint_use_stream(struct ::oslam_mappoint* h, hipStream_t s);

static inline int div_up(int a, int b) { return (a + b - 1) / b; }

// Device -> pinned-host transfer as a KERNEL on the caller's stream (the pinned block of hipHostMalloc is mapped into the device's address space: the stores go
// over PCIe), used for the small downloads that end an operator (local-BA results and control blocks, Fuse matches, status words): the copy stays in the stream's
// own queue instead of going through the runtime's copy path, which all streams of the process share.  Same-box A/B in the steady-state bench: 17.3 k frames/s
// against 17.0 k with hipMemcpyAsync (OSLAM_D2H_SDMA=1) — within the spread; kept because it removes a dependency on that shared path.  `dst` must be
// device-accessible host memory.
hipError_t copy_to_host_async(void* dst_pinned, const void* src_dev, size_t bytes, hipStream_t s);   // (defined once, in orb_extractor.hip, next to stream_wait)
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// Pinned host staging for the single-frame host-pointer entry points: uploads are memcpy'd into the pinned block and
// enqueued as async copies on the default stream, downloads land in the block and are handed out after ONE stream
// synchronisation.  (A pageable hipMemcpy costs ~15-25 us of host time each; the drop-in calls have 5-10 of them.)
struct PinStage {
    uint8_t* p = nullptr;
    size_t cap = 0, off = 0;
    void reset() { off = 0; }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; off = 0; }
    // reserve `bytes` (256-aligned); grows only while empty, so call reserve_total() first for a whole call
    int reserve_total(size_t bytes) {
        if (bytes <= cap) return OSLAM_OK;
        OSLAM_HIP_CHECK(hipDeviceSynchronize());
        release();
        const size_t ncap = bytes + bytes / 2 + 4096;
        OSLAM_HIP_CHECK(hipHostMalloc((void**)&p, ncap, 0));
        cap = ncap;
        return OSLAM_OK;
    }
    uint8_t* take(size_t bytes) { uint8_t* at = p + off; off += align_up(bytes ? bytes : 1, 256); return at; }
    int upload(void* dst, const void* src, size_t bytes) {
        if (!bytes) return OSLAM_OK;
        if (off + align_up(bytes, 256) > cap) { set_error("pinned staging overflow"); return OSLAM_E_CAPACITY; }
        uint8_t* at = take(bytes);
        memcpy(at, src, bytes);
        OSLAM_HIP_CHECK(hipMemcpyAsync(dst, at, bytes, hipMemcpyHostToDevice, nullptr));
        return OSLAM_OK;
    }
    // enqueue a download; returns the pinned address the data will be at after the stream has drained
    int download(const void* src, size_t bytes, uint8_t** at_out) {
        if (off + align_up(bytes ? bytes : 1, 256) > cap) { set_error("pinned staging overflow"); return OSLAM_E_CAPACITY; }
        uint8_t* at = take(bytes);
        if (bytes) OSLAM_HIP_CHECK(hipMemcpyAsync(at, src, bytes, hipMemcpyDeviceToHost, nullptr));
        *at_out = at;
        return OSLAM_OK;
    }
};

}  // namespace oslam
