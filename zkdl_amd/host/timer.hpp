// drop-in include name of the reference header timer.cuh / .hpp
#pragma once
#include "zkdl.hpp"
