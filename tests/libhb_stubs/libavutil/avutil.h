#include "libav_stub.h"
