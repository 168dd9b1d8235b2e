/* stand-in for <jansson.h>: see libav_stub.h */
#ifndef JANSSON_STUB_H
#define JANSSON_STUB_H
#include <stddef.h>
#include <stdio.h>
#include <stdarg.h>
typedef struct json_t json_t;
typedef long long json_int_t;
#endif
