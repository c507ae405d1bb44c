/* Prints the layout of the structs that cross the C-ABI (include/rocoder_hip.h) as JSON: field -> [offset, size].
 * tests/test_cabi_host.py compiles and runs it, and checks the output against tests/golden/abi_layout.json (the table
 * INTEGRATION.md shows next to the Rust #[repr(C)] structs) and against the ctypes mirror in rocoder_amd/_lib.py. */
#include <stddef.h>
#include <stdio.h>

#include "rocoder_hip.h"

#define F(T, f) printf("%s    \"%s\": [%zu, %zu]", first ? "" : ",\n", #f, offsetof(T, f), sizeof(((T *)0)->f)), first = 0
int main(void) {
    int first = 1;
    printf("{\n");
    printf("  \"rc_config\": {\n    \"sizeof\": [%zu, %zu]", sizeof(rc_config), _Alignof(rc_config));
    first = 0;
    F(rc_config, struct_size); F(rc_config, window_len); F(rc_config, factor); F(rc_config, amplitude);
    F(rc_config, pitch_multiple); F(rc_config, sample_rate); F(rc_config, channels); F(rc_config, buffer_secs);
    F(rc_config, seed); F(rc_config, device); F(rc_config, max_batch_hops); F(rc_config, window);
    F(rc_config, kernel); F(rc_config, kernel_user); F(rc_config, kernel_time_ms); F(rc_config, kernel_threads);
    F(rc_config, device_kernel); F(rc_config, dk_gain); F(rc_config, dk_gain_outside); F(rc_config, dk_lo_bin);
    F(rc_config, dk_hi_bin); F(rc_config, dk_shift_bins);
    printf("\n  },\n  \"rc_params\": {\n    \"sizeof\": [%zu, %zu]", sizeof(rc_params), _Alignof(rc_params));
    F(rc_params, window_len); F(rc_params, half_window_len); F(rc_params, samples_needed_per_window);
    F(rc_params, sample_step_len); F(rc_params, hops_per_window); F(rc_params, window_out_len);
    F(rc_params, corrected_amp_factor); F(rc_params, pitch_shifted_factor);
    printf("\n  },\n  \"rc_shard\": {\n    \"sizeof\": [%zu, %zu]", sizeof(rc_shard), _Alignof(rc_shard));
    F(rc_shard, device_index); F(rc_shard, ch_first); F(rc_shard, ch_count); F(rc_shard, win_first); F(rc_shard, win_count);
    printf("\n  }\n}\n");
    return 0;
}
