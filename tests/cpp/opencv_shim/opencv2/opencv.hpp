// TEST-ONLY: see core.hpp in this directory.
#pragma once
#include "core.hpp"
