// common.h -- integer shorthands of the reference's src/common.h:32-35, so that
// code written against libgwaspp's headers compiles against this mirror too.
#pragma once
#include <sys/types.h>  // glibc already provides ushort, uint, ulong

typedef unsigned char byte;
