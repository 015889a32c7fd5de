// drop-in include name of the reference header proof.cuh / .hpp
#pragma once
#include "zkdl.hpp"
