// abi.hip -- ABI self-description so the Python ctypes mirror can verify struct layouts at load.
#include "hip_compat.h"
#include "pase_amd.h"

extern "C" int pase_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(PaseConvGemm);
        case 1: return (int)sizeof(PaseWgrad);
        case 2: return (int)sizeof(PaseActBwd);
        case 3: return (int)sizeof(PaseAddBlocks);
        case 4: return (int)sizeof(PaseMlpHead1);
        default: return -1;
    }
}
