// drop-in include name of the reference header zkrelu.cuh / .hpp
#pragma once
#include "zkdl.hpp"
