/*
 * huffman/sys.h - error-routing helper macros under the names of the reference's
 * include/huffman/sys.h:10-78 (the reference's setup_ffi.py:30-43 includes this header; nothing of
 * it goes into cffi's cdef()).  A function body that uses them declares its error variable with
 * routine_m(), jumps to the label placed by routine_ensure_m() on a failed check, and leaves with
 * routine_defer_m().
 */
#ifndef INCLUDE_huffman_sys_h__
#define INCLUDE_huffman_sys_h__

#include <stdio.h>

#include "errors.h"

#define void_pptr_m(pointer) ((void**)(pointer))

/* declares the function's error variable */
#define routine_m() huf_error_t __error = HUF_ERROR_SUCCESS

/* the label every failed check jumps to */
#define routine_ensure_m() ensure:

/* return what the routine ended with */
#define routine_defer_m() do { return __error; } while (0)

/* label + return, for routines without clean-up code */
#define routine_yield_m() do { routine_ensure_m(); return __error; } while (0)

/* a zero / NULL parameter is HUF_ERROR_INVALID_ARGUMENT */
#define routine_param_m(param) \
    do { if ((param) == 0) { __error = HUF_ERROR_INVALID_ARGUMENT; goto ensure; } } while (0)

/* low <= value <= high, else HUF_ERROR_INVALID_ARGUMENT */
#define routine_inrange_m(value, low, high) \
    do { if ((value) < (low) || (value) > (high)) { __error = HUF_ERROR_INVALID_ARGUMENT; goto ensure; } } while (0)

/* leave with the given error */
#define routine_error_m(error) do { __error = (error); goto ensure; } while (0)

#define routine_success_m() routine_error_m(HUF_ERROR_SUCCESS)

/* true once the routine has been interrupted by an error */
#define routine_violation_m() (__error != HUF_ERROR_SUCCESS)

#endif /* INCLUDE_huffman_sys_h__ */
