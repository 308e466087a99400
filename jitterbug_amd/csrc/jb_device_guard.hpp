// jb_device_guard.hpp — "make the handle's device current for the duration of one entry point, then put the caller's back".
//
// A process may hold handles on several GPUs, or call torch.cuda.set_device after jb_create; every entry point that allocates
// or launches therefore runs with the HANDLE's device current.  It must not leave it current: torch (and any other HIP user of
// the thread) reads the current device through hipGetDevice, so an entry point that silently switched it would send the
// caller's next allocation or launch to the wrong GPU.  The guard is written against a tiny API trait so that tests/ can run
// it on the host with a recording stub (tests/host_harness.cpp); the library instantiates it with hipGetDevice / hipSetDevice.
#pragma once

namespace jb {

template <typename Api>            // Api::get(int*) / Api::set(int): 0 on success
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    // returns 0, or the Api's error code when the handle's device cannot be made current (nothing to restore then)
    int enter(int device) {
        if (Api::get(&prev) != 0) prev = -1;
        if (prev == device) return 0;
        const int rc = Api::set(device);
        if (rc != 0) return rc;
        switched = true;
        return 0;
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) Api::set(prev);        // every return path of the entry point, early error returns included
    }
    DeviceGuard() = default;
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

}  // namespace jb
