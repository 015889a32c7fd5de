// drop-in include name of the reference header zkfc.cuh / .hpp
#pragma once
#include "zkdl.hpp"
