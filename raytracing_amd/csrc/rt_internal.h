// rt_internal.h -- shared by the host-side translation units of librt_amd.so
#ifndef RT_INTERNAL_H
#define RT_INTERNAL_H
#include <stdarg.h>

// records the message returned by rt_last_error() for the calling thread and returns `status`
int rt_fail(int status, const char* fmt, ...);

#endif
