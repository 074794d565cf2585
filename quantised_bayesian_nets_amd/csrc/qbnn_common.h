// Shared by the translation units of libqbnn_hip.so: error reporting behind qbnn_last_error().
#ifndef QBNN_COMMON_H_
#define QBNN_COMMON_H_
#include <atomic>
#include <stdint.h>
#define QBNN_EXPORT extern "C" __attribute__((visibility("default")))
int qbnn_fail_msg(int code, const char* msg);       // records msg for qbnn_last_error(), returns code
int qbnn_check_launch_msg(const char* what);        // hipGetLastError() -> QBNN_OK / QBNN_E_LAUNCH
const unsigned int* qbnn_noise_dev();               // this thread's device noise source (qbnn_set_device_noise_source) or nullptr
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per device and kernel (`done` = the kernel's per-device bit mask)
int qbnn_ensure_dyn_lds(const void* fn, std::atomic<uint64_t>* done, int bytes);
#endif
