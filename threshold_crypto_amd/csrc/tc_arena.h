// Geometry and descriptor of the per-context table arena (tc_table.h explains what lives in it); shared by the
// device headers and the host-side launch prototypes.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace tc {

constexpr int kTblCoordWords = 16;   // 14 limbs + 2 words: one 64-byte row per coordinate and lane
constexpr int kTblEntryWords = 64;   // affine G2 entry: x (2 rows), y (2 rows); word 15 of a lane's x row = infinity flag
constexpr int kPairTableWords = 8 * kTblEntryWords;    // the 8-entry table of tc_gls.h
constexpr int kWaveTableWords = 32 * kPairTableWords;  // 64 KB per wave
constexpr int kG1EntryWords = 32;                      // affine G1 entry: x, y (14 limbs each) + 4 words: one 128-byte line
constexpr int kLaneTableWords = 8 * kG1EntryWords;     // the 8-entry base-4 GLV table of ONE lane: 1 KB (64 lanes: the same 64 KB slot)
constexpr uint32_t kTableXccs = 8;
constexpr uint32_t kSlotsPerXcc = 512;
constexpr size_t kTableArenaWords = (size_t)kTableXccs * kSlotsPerXcc * kWaveTableWords;  // 256 MB of int32
constexpr size_t kTableArenaFlags = (size_t)kTableXccs * kSlotsPerXcc;

// device pointers, owned by the context; passed by value to every kernel that may build a table
struct TableArena {
  int32_t* mem;     // kTableArenaWords words
  uint32_t* flags;  // kTableArenaFlags words, zeroed at creation; 1 = slot in use
};

}  // namespace tc
