// Pins dint/constants.hpp (the product's compile-time constants) to the REFERENCE's own configuration header,
// /root/reference/include/dint/dint_configuration.hpp, compiled where it lies (-I points at it; it needs <cmath> and
// <limits> only). Every static_assert below is checked when `make -C oracle ref` runs in a container that has the
// reference tree; the values are also exported so that a test can compare them with tests/golden/ref_constants.json on
// machines without it. This file contains no reference code.
#include <cstdint>

#include "dint_configuration.hpp"  // the reference's
#include "dint/constants.hpp"      // the product's (-I dint_amd/csrc/host)

static_assert(EXCEPTIONS == dint::kExceptions, "dint_configuration.hpp:6");
static_assert(ds2i::constants::num_selectors == dint::kNumSelectors, "dint_configuration.hpp:20");
static_assert(ds2i::constants::max_entry_size == dint::kMaxEntrySize, "dint_configuration.hpp:24");
static_assert(ds2i::constants::num_entries == dint::kNumEntries, "dint_configuration.hpp:26");
static_assert(ds2i::constants::log2_num_entries == 16 && (1u << ds2i::constants::log2_num_entries) == dint::kNumEntries,
              "dint_configuration.hpp:27");
static_assert(sizeof(ds2i::constants::target_sizes) / sizeof(uint32_t) == dint::kNumTargetSizes, "dint_configuration.hpp:25");
static_assert(sizeof(ds2i::constants::selector_codes) / sizeof(uint32_t) == dint::kNumSelectors, "dint_configuration.hpp:21");
static_assert(ds2i::constants::context == ds2i::constants::block_selector::max, "dint_configuration.hpp:19: context MAX");
static_assert(dint::kReserved == EXCEPTIONS + 5, "single_dictionary.hpp:22: reserved = EXCEPTIONS + 5");

// target_sizes / selector_codes are `static const` arrays (not constexpr): compared at load time
extern "C" int ref_constants_check(void) {
    for (uint32_t i = 0; i != dint::kNumTargetSizes; ++i)
        if (ds2i::constants::target_sizes[i] != dint::kTargetSizes[i]) return 1;
    for (uint32_t i = 0; i != dint::kNumSelectors; ++i)
        if (ds2i::constants::selector_codes[i] != i) return 2;
    if (ds2i::constants::num_target_sizes != dint::kNumTargetSizes) return 3;  // std::log2(max_entry_size) + 1, :28
    return 0;
}
// the reference's values, for tests/golden/ref_constants.json
extern "C" int ref_constants(uint32_t* out, int cap) {
    const uint32_t v[] = {EXCEPTIONS, ds2i::constants::num_selectors, ds2i::constants::max_entry_size,
                          ds2i::constants::num_entries, ds2i::constants::log2_num_entries, ds2i::constants::num_target_sizes,
                          ds2i::constants::target_sizes[0], ds2i::constants::target_sizes[1], ds2i::constants::target_sizes[2],
                          ds2i::constants::target_sizes[3], ds2i::constants::target_sizes[4]};
    const int n = int(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i != n && i != cap; ++i) out[i] = v[i];
    return n;
}
