// drop-in include name of the reference header g1-tensor.cuh / .hpp
#pragma once
#include "zkdl.hpp"
