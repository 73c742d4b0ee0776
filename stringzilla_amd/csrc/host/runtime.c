/*
 *  runtime.c - version, capabilities, unified allocator, device scopes, status strings, grow-only buffers.
 *
 *  ROCm counterpart of the reference's c/stringzillas/runtime.cuh (226 lines) and of the error plumbing in
 *  c/stringzillas/stringzillas.cuh:207-257.  Same entry points, same status conventions; HIP instead of the
 *  CUDA driver API; no CPU executors (this build ships GPU engines only).
 */
#include "szs_internal.h"

#include <stdlib.h>
#include <string.h>

/* ---- status strings ---------------------------------------------------------------------------------------------- */

static char const *szs_message_for(sz_status_t status) {
    /* Same wording as the reference where a caller could be matching on it (stringzillas.cuh:218-231). */
    switch (status) {
    case sz_success_k: return NULL;
    case sz_bad_alloc_k: return "Memory allocation failed";
    case sz_invalid_utf8_k: return "Invalid UTF-8 input";
    case sz_contains_duplicates_k: return "Input contains duplicates";
    case sz_overflow_risk_k: return "Overflow risk detected";
    case sz_unexpected_dimensions_k: return "Input/output size mismatch";
    case sz_missing_gpu_k: return "GPU device not available or HIP not initialized";
    case sz_device_code_mismatch_k: return "Backend and executor mismatch";
    case sz_device_memory_mismatch_k: return "Use device-reachable or unified memory";
    case sz_status_unknown_k: return "Unknown error";
    default: return "Unrecognized error code";
    }
}

sz_status_t szs_report(sz_status_t status, char const **error_message, char const *override_message) {
    if (error_message) *error_message = override_message ? override_message : szs_message_for(status);
    return status;
}

sz_status_t szs_report_hip(hipError_t error, char const **error_message) {
    if (error == hipSuccess) return szs_report(sz_success_k, error_message, NULL);
    /* Like the reference, hand back the stable runtime error *name* (stringzillas.cuh:236-256). */
    sz_status_t const status = error == hipErrorOutOfMemory   ? sz_bad_alloc_k
                               : error == hipErrorNoDevice    ? sz_missing_gpu_k
                               : error == hipErrorInvalidDevice ? sz_missing_gpu_k
                                                              : sz_status_unknown_k;
    (void)hipGetLastError(); /* clear the sticky error so the next call starts clean */
    return szs_report(status, error_message, hipGetErrorName(error));
}

/* ---- version & capabilities (runtime.cuh:15-56) ------------------------------------------------------------------ */

int szs_version_major(void) { return SZS_VERSION_MAJOR; }
int szs_version_minor(void) { return SZS_VERSION_MINOR; }
int szs_version_patch(void) { return SZS_VERSION_PATCH; }

sz_capability_t szs_capabilities_comptime(void) { return (sz_capability_t)(sz_cap_serial_k | sz_cap_cuda_k); }

sz_capability_t szs_capabilities_runtime(void) {
    /* A GPU that cannot be enumerated, for whatever reason, contributes no bits (runtime.cuh:37-45). */
    int devices = 0;
    hipError_t const error = hipGetDeviceCount(&devices);
    if (error != hipSuccess) (void)hipGetLastError();
    unsigned caps = sz_cap_serial_k;
    if (error == hipSuccess && devices > 0) caps |= sz_cap_cuda_k;
    return (sz_capability_t)caps;
}

sz_capability_t szs_capabilities(void) {
    static sz_capability_t cached = sz_caps_none_k; /* both sides carry `serial`, so 0 doubles as "not probed" */
    if (cached == sz_caps_none_k) cached = (sz_capability_t)(szs_capabilities_comptime() & szs_capabilities_runtime());
    return cached;
}

/* ---- unified memory (runtime.cuh:58-69,205-222; types.cuh:145-151) ------------------------------------------------ */

void *szs_unified_alloc(sz_size_t size_bytes) {
    void *pointer = NULL;
    if (!size_bytes) size_bytes = 1;
    if (hipMallocManaged(&pointer, size_bytes, hipMemAttachGlobal) != hipSuccess) {
        (void)hipGetLastError();
        return NULL;
    }
    return pointer;
}

void szs_unified_free(void *pointer, sz_size_t size_bytes) {
    (void)size_bytes;
    if (pointer) (void)hipFree(pointer);
}

static void *szs_unified_allocate_thunk(sz_size_t size_bytes, void *handle) {
    (void)handle;
    return szs_unified_alloc(size_bytes);
}
static void szs_unified_free_thunk(void *pointer, sz_size_t size_bytes, void *handle) {
    (void)handle;
    szs_unified_free(pointer, size_bytes);
}

sz_status_t sz_memory_allocator_init_unified(sz_memory_allocator_t *alloc, char const **error_message) {
    if (!alloc) return szs_report(sz_status_unknown_k, error_message, "Allocator must not be null");
    alloc->allocate = &szs_unified_allocate_thunk;
    alloc->free = &szs_unified_free_thunk;
    alloc->handle = NULL;
    return szs_report(sz_success_k, error_message, NULL);
}

/* ---- device scopes (runtime.cuh:74-201) --------------------------------------------------------------------------- */

static sz_status_t szs_scope_new(szs_scope_kind_t kind, size_t cpu_cores, int gpu_device, szs_device_scope_t *out,
                                 char const **error_message) {
    if (!out) return szs_report(sz_status_unknown_k, error_message, "Scope must not be null");
    szs_scope_s *scope = (szs_scope_s *)calloc(1, sizeof(szs_scope_s));
    if (!scope) return szs_report(sz_bad_alloc_k, error_message, NULL);
    scope->kind = kind, scope->cpu_cores = cpu_cores, scope->gpu_device = gpu_device, scope->stream = NULL;
    *out = scope;
    return szs_report(sz_success_k, error_message, NULL);
}

sz_status_t szs_device_scope_init_default(szs_device_scope_t *scope, char const **error_message) {
    return szs_scope_new(szs_scope_default_k, 1, 0, scope, error_message);
}

sz_status_t szs_device_scope_init_cpu_cores(sz_size_t cpu_cores, szs_device_scope_t *scope, char const **error_message) {
    /* One core folds back onto the default scope, as in the reference (runtime.cuh:87-95). */
    if (cpu_cores == 1) return szs_scope_new(szs_scope_default_k, 1, 0, scope, error_message);
    return szs_scope_new(szs_scope_cpu_k, cpu_cores, -1, scope, error_message);
}

sz_status_t szs_device_scope_init_gpu_device(sz_size_t gpu_device, szs_device_scope_t *scope, char const **error_message) {
    int devices = 0;
    hipError_t const error = hipGetDeviceCount(&devices);
    if (error != hipSuccess) {
        (void)hipGetLastError();
        return szs_report(sz_missing_gpu_k, error_message, NULL);
    }
    if (gpu_device >= (sz_size_t)devices) return szs_report(sz_missing_gpu_k, error_message, NULL);
    return szs_scope_new(szs_scope_gpu_k, 0, (int)gpu_device, scope, error_message);
}

sz_status_t szs_device_scope_get_cpu_cores(szs_device_scope_t handle, sz_size_t *cpu_cores, char const **error_message) {
    szs_scope_s *scope = (szs_scope_s *)handle;
    if (!scope || !cpu_cores) return szs_report(sz_status_unknown_k, error_message, "Scope must not be null");
    if (scope->kind == szs_scope_gpu_k) return szs_report(sz_status_unknown_k, error_message, "Not a CPU scope");
    *cpu_cores = scope->cpu_cores;
    return szs_report(sz_success_k, error_message, NULL);
}

sz_status_t szs_device_scope_get_gpu_device(szs_device_scope_t handle, sz_size_t *gpu_device, char const **error_message) {
    szs_scope_s *scope = (szs_scope_s *)handle;
    if (!scope || !gpu_device) return szs_report(sz_status_unknown_k, error_message, "Scope must not be null");
    if (scope->kind != szs_scope_gpu_k) return szs_report(sz_status_unknown_k, error_message, "Not a GPU scope");
    *gpu_device = (sz_size_t)scope->gpu_device;
    return szs_report(sz_success_k, error_message, NULL);
}

sz_status_t szs_device_scope_get_capabilities(szs_device_scope_t handle, sz_capability_t *capabilities,
                                              char const **error_message) {
    szs_scope_s *scope = (szs_scope_s *)handle;
    if (!scope || !capabilities) return szs_report(sz_status_unknown_k, error_message, "Scope must not be null");
    /* GPU scopes answer with the GPU bits, CPU scopes with the CPU bits (runtime.cuh:179-201).  In this build the
     * default scope lazily binds GPU 0 for every engine, so it reports the union: bindings that infer capabilities
     * from the default scope then select the (only) GPU engines. */
    sz_capability_t const system = szs_capabilities();
    if (scope->kind == szs_scope_gpu_k) *capabilities = (sz_capability_t)(system & sz_caps_cuda_k);
    else if (scope->kind == szs_scope_cpu_k) *capabilities = (sz_capability_t)(system & sz_caps_cpus_k);
    else *capabilities = system;
    return szs_report(sz_success_k, error_message, NULL);
}

void szs_device_scope_free(szs_device_scope_t handle) {
    szs_scope_s *scope = (szs_scope_s *)handle;
    if (!scope) return;
    if (scope->stream) {
        int previous = 0;
        (void)hipGetDevice(&previous);
        (void)hipSetDevice(scope->gpu_device);
        (void)hipStreamDestroy(scope->stream);
        (void)hipSetDevice(previous);
    }
    free(scope);
}

sz_status_t szs_scope_bind_gpu(szs_scope_s *scope, int *device, hipStream_t *stream, char const **error_message) {
    if (!scope) return szs_report(sz_status_unknown_k, error_message, "Scope must not be null");
    int const cpu_requests_on_gpu = szs_tuning_get(szs_knob_cpu_requests_k) == 1; /* see engines.c: engine_new */
    if (scope->kind == szs_scope_cpu_k && !cpu_requests_on_gpu) return szs_report(sz_device_code_mismatch_k, error_message, NULL);
    if (scope->kind == szs_scope_default_k || scope->kind == szs_scope_cpu_k) {
        /* A default scope used with a GPU engine binds device 0 (stringzillas.cuh:303-320,355-365). */
        int devices = 0;
        if (hipGetDeviceCount(&devices) != hipSuccess || devices <= 0) {
            (void)hipGetLastError();
            return szs_report(sz_missing_gpu_k, error_message, NULL);
        }
        scope->gpu_device = 0;
    }
    hipError_t error = hipSetDevice(scope->gpu_device); /* made current per calling thread, every call */
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    if (!scope->stream) {
        error = hipStreamCreateWithFlags(&scope->stream, hipStreamNonBlocking);
        if (error != hipSuccess) return szs_report_hip(error, error_message);
    }
    *device = scope->gpu_device;
    *stream = scope->stream;
    return szs_report(sz_success_k, error_message, NULL);
}

/* ---- grow-only buffers --------------------------------------------------------------------------------------------- */

void szs_buffer_release(szs_buffer_t *buffer) {
    if (!buffer->pointer) return;
    switch (buffer->kind) {
    case szs_memory_host_k: free(buffer->pointer); break;
    case szs_memory_pinned_k: (void)hipHostFree(buffer->pointer); break;
    case szs_memory_device_k: (void)hipFree(buffer->pointer); break;
    }
    buffer->pointer = NULL, buffer->capacity = 0;
}

sz_status_t szs_buffer_reserve(szs_buffer_t *buffer, szs_memory_kind_t kind, int device, size_t bytes,
                               char const **error_message) {
    if (buffer->pointer && buffer->kind == kind && buffer->capacity >= bytes &&
        (kind == szs_memory_host_k || buffer->device == device))
        return sz_success_k;
    szs_buffer_release(buffer);
    size_t capacity = 4096;
    while (capacity < bytes) capacity += capacity / 2 + 4096; /* geometric growth, like the reference's safe_vector */
    void *pointer = NULL;
    hipError_t error = hipSuccess;
    switch (kind) {
    case szs_memory_host_k: pointer = malloc(capacity); break;
    case szs_memory_pinned_k: error = hipHostMalloc(&pointer, capacity, hipHostMallocDefault); break;
    case szs_memory_device_k: error = hipMalloc(&pointer, capacity); break;
    }
    if (error != hipSuccess) return szs_report_hip(error, error_message);
    if (!pointer) return szs_report(sz_bad_alloc_k, error_message, NULL);
    buffer->pointer = pointer, buffer->capacity = capacity, buffer->kind = kind, buffer->device = device;
    return sz_success_k;
}
