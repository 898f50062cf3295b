// Build shim: ASCII/pass-through stand-in for utf8proc (reference use: util/string.cpp:193-200).
#pragma once
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <sys/types.h>
typedef uint8_t utf8proc_uint8_t;
typedef ssize_t utf8proc_ssize_t;
typedef int utf8proc_option_t;
#define UTF8PROC_COMPOSE 1
static inline utf8proc_ssize_t utf8proc_map(const utf8proc_uint8_t *s, utf8proc_ssize_t len,
                                            utf8proc_uint8_t **dst, utf8proc_option_t) {
    *dst = (utf8proc_uint8_t *)malloc(len + 1);
    memcpy(*dst, s, len);
    (*dst)[len] = 0;
    return len;
}
static inline const char *utf8proc_errmsg(utf8proc_ssize_t) { return "utf8proc shim error"; }
