// Lets g++ read the device-side field / permutation headers of plonky2_goldibear_amd/csrc (which are plain integer C++ apart from
// their qualifiers and a few instruction-level helpers that carry a portable branch) so that tests/test_device_headers_on_host.py
// can run them on the CPU against the host mirrors.  Test infrastructure only.
#pragma once
#define __host__
#define __device__
#define __forceinline__ inline
#define __restrict__
