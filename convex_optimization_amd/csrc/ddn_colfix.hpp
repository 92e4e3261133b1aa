// Tile-generic column kernels with compile-time transforms for common image heights (float; the resolvent and the
// multiplier mode, the two that run once per solver iteration).  Included by one unit per mode, which defines
// DDN_COLFIX_MODE / DDN_COLFIX_NAME and, for the transform-only stages of a split column that go with that mode,
// DDN_COLFIX_STAGE_MODE / DDN_COLFIX_STAGE_NAME.
#pragma once
#include "ddn_launch.hpp"

// Heights whose columns are one transform and that the register-resident column kernels do not take; 50 and 60: the
// long factor B of heights that are split (2400 = 48 x 50, 3600 = 60 x 60, ...; 64 has its kernels in ddn_api.hip); 56 and 80:
// 3024 = 54 x 56 and 4000 = 50 x 80, the 12-megapixel photo formats 4032 x 3024 and 4000 x 3000 either way up.
#define DDN_COLH_FIXED(X) X(50) X(56) X(60) X(72) X(80) X(96) X(100) X(480) X(600) X(640) X(720) X(768) X(800) X(900) X(960) X(1000) X(1080) X(1200) \
    X(1280) X(1440) X(1536) X(1600) X(1920) X(2000) X(2160) X(2400)
// The short factor A of split heights (2560 = 40 x 64, 3072 = 48 x 64, 3200 = 50 x 64, 3840 = 60 x 64, ...): the
// transform-only stages either side of the middle one.
#define DDN_COLA_FIXED(X) X(40) X(48) X(50) X(54) X(60) X(63) X(75) X(80)

const void* DDN_COLFIX_NAME(int n) {
    switch (n) {
#define DDN_X(h) \
    case h: return (const void*)k_col<float, DDN_COLFIX_MODE, h>;
        DDN_COLH_FIXED(DDN_X)
#undef DDN_X
        default: return nullptr;
    }
}

const void* DDN_COLFIX_STAGE_NAME(int n) {
    switch (n) {
#define DDN_X(h) \
    case h: return (const void*)k_col<float, DDN_COLFIX_STAGE_MODE, h>;
        DDN_COLA_FIXED(DDN_X)
#undef DDN_X
        default: return nullptr;
    }
}
