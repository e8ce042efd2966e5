// Stand-in for <codec2/codec2.h> (build check only, see ../README.md): the three calls apps/m17-demod.cpp makes.
#pragma once
#include <cstring>
struct CODEC2 { int mode; };
#define CODEC2_MODE_3200 0
inline struct CODEC2* codec2_create(int mode) { static struct CODEC2 c; c.mode = mode; return &c; }
inline void codec2_destroy(struct CODEC2*) {}
inline void codec2_decode(struct CODEC2*, short* speech_out, const unsigned char*) { std::memset(speech_out, 0, 160 * sizeof(short)); }
