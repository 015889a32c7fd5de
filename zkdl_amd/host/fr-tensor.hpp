// drop-in include name of the reference header fr-tensor.cuh / .hpp
#pragma once
#include "zkdl.hpp"
