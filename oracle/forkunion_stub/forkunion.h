/**
 *  Declaration-only stand-in for the ForkUnion C API (an un-vendored git submodule of the reference:
 *  /root/reference/.gitmodules, /root/reference/forkunion is empty).
 *
 *  TEST INFRASTRUCTURE ONLY. The reference header include/stringzillas/types.hpp:15 includes <forkunion.h>
 *  and wraps these calls in `forkunion_executor_t` (types.hpp:156-258). The oracle shim never instantiates
 *  that executor (it shards rows over std::thread itself), so none of these symbols is ever linked.
 */
#ifndef SZS_ORACLE_FORKUNION_STUB_H_
#define SZS_ORACLE_FORKUNION_STUB_H_
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef void *fu_pool_t;
typedef void *fu_topology_t;
typedef void *fu_lambda_context_t;
typedef int fu_bool_t;
typedef enum { fu_capabilities_all_k = 0x7fffffff } fu_capabilities_t;
typedef enum { fu_caller_inclusive_k = 0, fu_caller_exclusive_k = 1 } fu_caller_exclusivity_t;
typedef void (*fu_for_prongs_t)(fu_lambda_context_t, size_t, size_t, size_t);
typedef void (*fu_for_slices_t)(fu_lambda_context_t, size_t, size_t, size_t, size_t);
typedef void (*fu_for_threads_t)(fu_lambda_context_t, size_t, size_t);
fu_topology_t fu_topology_new(void);
void fu_topology_delete(fu_topology_t);
size_t fu_logical_cores_count(fu_topology_t);
fu_pool_t fu_pool_new(char const *, fu_capabilities_t);
void fu_pool_delete(fu_pool_t);
fu_bool_t fu_pool_spawn(fu_topology_t, fu_pool_t, size_t, fu_caller_exclusivity_t);
size_t fu_pool_threads_count(fu_pool_t);
void fu_pool_for_n(fu_pool_t, size_t, fu_for_prongs_t, fu_lambda_context_t);
void fu_pool_for_n_dynamic(fu_pool_t, size_t, fu_for_prongs_t, fu_lambda_context_t);
void fu_pool_for_slices(fu_pool_t, size_t, fu_for_slices_t, fu_lambda_context_t);
void fu_pool_for_threads(fu_pool_t, fu_for_threads_t, fu_lambda_context_t);
#ifdef __cplusplus
}
#endif
#endif
