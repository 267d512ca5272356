// Compile-time constants of the DINT codec family.
//
// Mirrors the reference's compile-time configuration so that encoded bytes and
// dictionary files are interchangeable with it:
//   include/dint/dint_configuration.hpp:6,20,24-28  (EXCEPTIONS, num_selectors,
//       max_entry_size, target_sizes, num_entries)
//   include/util.hpp:33-35                          (min_size, max_size, block_size)
//   include/dint/single_dictionary.hpp:22           (reserved = EXCEPTIONS + 5)
#pragma once
#include <cstddef>
#include <cstdint>

namespace dint {

constexpr uint32_t kExceptions = 2;        // codewords 0 (16-bit) and 1 (32-bit)
constexpr uint32_t kNumRuns = 5;           // codewords 2..6: 256,128,64,32,16 zeros
constexpr uint32_t kReserved = kExceptions + kNumRuns;  // = 7
constexpr uint32_t kNumSelectors = 6;      // contexts of the multi dictionary
constexpr uint32_t kMaxEntrySize = 16;     // l
constexpr uint32_t kNumEntries = 65536;    // 2^b, b = 16
constexpr uint32_t kNumTargetSizes = 5;
constexpr uint32_t kTargetSizes[kNumTargetSizes] = {16, 8, 4, 2, 1};
constexpr uint32_t kBlockSize = 256;       // posting-list block / multi context block
constexpr uint64_t kMinListSize = 0;
constexpr uint64_t kMaxListSize = 50000000;
constexpr uint32_t kInvalidIndex = uint32_t(-1);

// size of run codeword `index` (2..6) -> 256 >> (index - 2)
constexpr uint32_t run_size_of(uint32_t index) { return 256u >> (index - kExceptions); }

enum class dict_kind : int {
    rectangular = 0,    // single_dictionary_rectangular_type  (dictionary_types.hpp:8-9)
    single_packed = 1,  // single_dictionary_packed_type       (dictionary_types.hpp:10-12)
    multi_packed = 2,   // multi_dictionary_packed_type        (dictionary_types.hpp:19-21)
};

}  // namespace dint
