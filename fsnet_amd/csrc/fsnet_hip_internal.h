#pragma once
#include "../../include/fsnet_hip.h"
