// libgwaspp.h -- word-size contract of the reference (src/libgwaspp/libgwaspp.h:44-45,
// 91-93; libgwasppConfig.h:8-13): DataBlock = uint16, PWORD = uint64, 4 blocks
// per processor word.
#pragma once
#include "common.h"

typedef ushort DataBlock;
typedef ulong PWORD;
enum { BLOCKS_PER_PWORD = 4, PROCESSOR_WORD_SIZE = 64 };
