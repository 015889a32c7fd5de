// drop-in include name of the reference header commitment.cuh / .hpp
#pragma once
#include "zkdl.hpp"
