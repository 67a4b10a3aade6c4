"""TEST INFRASTRUCTURE, CPU only: compile flags of the checking builds of the host emulation (tests/emu_solver.py builds them, tests/test_emu_sanitize.py runs them).
Not shipped to the GPU box (.gpurunignore): the sanitizers run on the CPU build of the kernel source only."""
VARIANTS = {"race": ("libobca_emu_race.so", ["-O0", "-g", "-fno-omit-frame-pointer", "-DOBCA_EMU_RACE", "-Wno-frame-address"]),      # cross-lane hazards through HBM
            "ubsan": ("libobca_emu_ubsan.so", ["-O1", "-g", "-fsanitize=undefined,bounds-strict", "-fno-sanitize-recover=undefined"]),      # index / shift / overflow checks
            "asan": ("libobca_emu_asan.so", ["-O1", "-g", "-fno-omit-frame-pointer", "-DOBCA_EMU_ASAN", "-fsanitize=address"])}      # exact buffer sizes under AddressSanitizer
